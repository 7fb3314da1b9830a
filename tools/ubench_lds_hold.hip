// Does the LDS of a workgroup that spins (s_sleep) survive while other queues of the process keep the device busy?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_lds_hold.hip -o tools/ubench_lds_hold && tools/ubench_lds_hold [seconds]
// Three streams, each launching `hold` again and again: workgroups of 512 threads with 80 KB of LDS (two per CU) fill
// their LDS with a pattern, spin for a while as a chain pair does while it waits for its predecessor, and check the
// pattern; two more streams stream through memory meanwhile.  Prints the number of words found changed.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(512) hold(int lds_bytes, long long spin, int rounds, unsigned long long *bad, int *first) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned *L = reinterpret_cast<unsigned *>(smem);
  const int n = lds_bytes / 4;
  for (int r = 0; r < rounds; ++r) {
    const unsigned key = 0x9E3779B9u * (blockIdx.x * 131u + r + 1u);
    for (int i = threadIdx.x; i < n; i += blockDim.x) L[i] = key ^ (unsigned)i;
    __syncthreads();
    if (threadIdx.x == 0) {                                    // one lane polls, as wait_for does
      long long t0 = wall_clock64();
      while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    int wrong = 0, at = 1 << 30;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
      if (L[i] != (key ^ (unsigned)i)) {
        ++wrong;
        at = i < at ? i : at;
      }
    if (wrong) {
      atomicAdd(bad, (unsigned long long)wrong);
      atomicMin(first, at * 4);
    }
    __syncthreads();
  }
}

__global__ void stream_copy(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main(int argc, char **argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
  const int lds = 80 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void *>(hold), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  unsigned long long *bad;
  int *first;
  hipMalloc(&bad, 8);
  hipMalloc(&first, 4);
  hipMemset(bad, 0, 8);
  int big = 1 << 30;
  hipMemcpy(first, &big, 4, hipMemcpyHostToDevice);
  const size_t n = (size_t)512 << 20;                          // 512 MB of float4 per buffer
  float4 *a, *b, *c, *d;
  hipMalloc(&a, n), hipMalloc(&b, n), hipMalloc(&c, n), hipMalloc(&d, n);
  hipMemset(a, 1, n), hipMemset(c, 2, n);
  hipStream_t s[5];
  for (auto &x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    for (int q = 0; q < 3; ++q) hipLaunchKernelGGL(hold, dim3(1280), dim3(512), lds, s[q], lds, 20000ll, 4, bad, first);   // 200 us spins
    hipLaunchKernelGGL(stream_copy, dim3(4096), dim3(256), 0, s[3], a, b, n / 16);
    hipLaunchKernelGGL(stream_copy, dim3(4096), dim3(256), 0, s[4], c, d, n / 16);
    launches += 3;
    if (launches % 30 == 0) hipDeviceSynchronize();
  }
  hipDeviceSynchronize();
  unsigned long long h;
  int f;
  hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
  hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
  printf("%ld launches of 1280 workgroups x 4 rounds (80 KB of LDS held over a 200 us spin, three queues + two streaming queues): "
         "%llu words found changed", launches, h);
  if (h) printf(", lowest byte offset %d", f);
  printf("\n");
  return 0;
}
