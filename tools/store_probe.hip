// Probe: how fast can ONE CU issue / complete 128 KiB of 16-byte-per-lane full-line stores, alone and with every CU
// doing the same (lockstep vs de-phased)?  Decides whether the GEMM epilogue's store tail is a per-CU or a chip limit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(2);} } while (0)

__global__ __launch_bounds__(512) void probe(uint4* out, int iters, int dephase, long long* t_issue, long long* t_done,
                                             size_t region_chunks) {
  __shared__ char pad[100 * 1024];   // 1 workgroup per CU
  pad[threadIdx.x] = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (dephase && (blockIdx.x >> 3 & 1)) for (int z = 0; z < dephase; ++z) __builtin_amdgcn_s_sleep(127);
  long long ti = 0, td = 0;
  for (int it = 0; it < iters; ++it) {
    // ~12 us of "compute"
    for (int z = 0; z < 3; ++z) __builtin_amdgcn_s_sleep(127);
    __syncthreads();
    const size_t base = ((size_t)it * gridDim.x + blockIdx.x) * 8192 % region_chunks;   // 128 KiB = 8192 x 16 B
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      // wave tile: 128 rows x 128 B; instruction k covers 8 rows
      out[base + (size_t)wave * 1024 + k * 64 + lane] = make_uint4(it, k, lane, wave);
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = __builtin_readcyclecounter();
    ti += t1 - t0; td += t2 - t0;
  }
  if (lane == 0) { atomicAdd((unsigned long long*)t_issue, (unsigned long long)ti); atomicAdd((unsigned long long*)t_done, (unsigned long long)td); }
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  uint4* buf; CK(hipMalloc(&buf, bytes));
  long long *ti, *td; CK(hipMalloc(&ti, 8)); CK(hipMalloc(&td, 8));
  const int iters = 20;
  for (int dephase : {0, 2}) for (int grid : {1, 8, 64, 256}) {
    CK(hipMemset(ti, 0, 8)); CK(hipMemset(td, 0, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    probe<<<grid, 512>>>(buf, iters, dephase, ti, td, bytes / 16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long hi, hd; CK(hipMemcpy(&hi, ti, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hd, td, 8, hipMemcpyDeviceToHost));
    const double n = (double)grid * 8 * iters;
    printf("grid %3d dephase %d: issue %.0f cyc, issue+drain %.0f cyc per 128 KiB (s_memtime ticks @100MHz?), kernel %.3f ms (%.1f us/iter)\n",
           grid, dephase, hi / n, hd / n, ms, ms * 1e3 / iters);
  }
  return 0;
}
