// Does a DS access beyond a workgroup's LDS allocation reach the neighbour workgroup on the same CU?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_lds_oob.hip -o /tmp/lds_oob && /tmp/lds_oob
// Two workgroups of 512 threads with 80 KB of dynamic LDS each share a CU.  Every workgroup fills its LDS with its own
// id, waits, then thread 0 stores a marker `beyond` bytes past its end, waits again, and everybody scans the own LDS for
// foreign values.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(512) k(int lds_bytes, int beyond, int *found, int *where, long long spin) {
  extern __shared__ __align__(16) unsigned char smem[];
  int *L = reinterpret_cast<int *>(smem);
  const int n = lds_bytes / 4, me = 0x1000000 + (int)blockIdx.x;
  for (int i = threadIdx.x; i < n; i += blockDim.x) L[i] = me;
  __syncthreads();
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);          // the neighbour fills its own meanwhile
  __syncthreads();
  if (threadIdx.x == 0) {
    volatile int *p = reinterpret_cast<volatile int *>(smem + lds_bytes + beyond);
    *p = 0x7E57;                                                           // beyond the allocation
    volatile int *q = reinterpret_cast<volatile int *>(smem + lds_bytes + beyond + 64);
    atomicOr((int *)q, 0x40000000);
  }
  __syncthreads();
  t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    if (L[i] != me) {
      atomicAdd(found, 1);
      atomicMin(where, i * 4);
    }
}

int main() {
  int *found, *where;
  hipMalloc(&found, 4);
  hipMalloc(&where, 4);
  const int lds = 80 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int beyond : {0, 64, 512, 4096}) {
    int zero = 0, big = 1 << 30;
    hipMemcpy(found, &zero, 4, hipMemcpyHostToDevice);
    hipMemcpy(where, &big, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(512), dim3(512), lds, 0, lds, beyond, found, where, 2000000ll);
    hipDeviceSynchronize();
    int f, w;
    hipMemcpy(&f, found, 4, hipMemcpyDeviceToHost);
    hipMemcpy(&w, where, 4, hipMemcpyDeviceToHost);
    printf("store %d bytes beyond the end of an 80 KB allocation (2 workgroups per CU): %d foreign words found in some workgroup's LDS%s\n",
           beyond, f, f ? "" : " (none)");
    if (f) printf("   lowest byte offset of a foreign word: %d\n", w);
  }
  return 0;
}
