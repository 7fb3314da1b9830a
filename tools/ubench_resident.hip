// How many workgroups of a given shape are on the device at once?  Every workgroup notes when it starts, spins for
// a fixed time and notes when it ends; the host counts overlaps.   hipcc --offload-arch=gfx950 -O2 tools/ubench_resident.hip -o tools/ubench_resident
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_spin(long long *t, long long ticks, int vgpr_ballast) {
  extern __shared__ unsigned char smem[];
  if (threadIdx.x == 0) {
    smem[0] = 1;
    long long t0 = wall_clock64();
    t[2 * blockIdx.x] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    t[2 * blockIdx.x + 1] = wall_clock64();
  }
  __syncthreads();
}

int main(int argc, char **argv) {
  int nt = argc > 1 ? atoi(argv[1]) : 1024, lds_kb = argc > 2 ? atoi(argv[2]) : 160, blocks = argc > 3 ? atoi(argv[3]) : 2048;
  int us = argc > 4 ? atoi(argv[4]) : 100;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s: %d CUs, %zu B LDS per workgroup max\n", p.name, p.multiProcessorCount, p.sharedMemPerBlock);
  long long *d;
  hipMalloc(&d, sizeof(long long) * 2 * blocks);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_spin), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(nt), lds_kb * 1024, 0, d, (long long)us * 100, 0);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("error %s\n", hipGetErrorString(e)); return 1; }
  }
  std::vector<long long> t(2 * blocks);
  hipMemcpy(t.data(), d, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
  std::vector<std::pair<long long, int>> ev;
  long long t0 = t[0];
  for (int i = 0; i < blocks; ++i) t0 = std::min(t0, t[2 * i]);
  for (int i = 0; i < blocks; ++i) { ev.push_back({t[2 * i], 1}); ev.push_back({t[2 * i + 1], -1}); }
  std::sort(ev.begin(), ev.end());
  int cur = 0, peak = 0;
  long long area = 0, last = t0, end = t0;
  for (auto &e : ev) { area += (long long)cur * (e.first - last); last = e.first; cur += e.second; peak = std::max(peak, cur); end = e.first; }
  printf("%d threads, %d KB LDS, %d workgroups of %d us: peak %d on the device, average %.1f, span %.1f us (ideal at %d CUs: %.1f us)\n", nt,
         lds_kb, blocks, us, peak, (double)area / (end - t0), (end - t0) / 100.0, p.multiProcessorCount,
         (double)blocks * us / p.multiProcessorCount);
  return 0;
}
