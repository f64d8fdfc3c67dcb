// micro-benchmark: throughput of global atomics on one address from many waves (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_ret(unsigned *ctr, unsigned *sink, int m) {
  unsigned acc = 0;
  for (int i = 0; i < m; ++i) { unsigned v = 0; if (threadIdx.x == 0) v = atomicAdd(ctr, 1u); acc += __shfl(v, 0, 64); }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
__global__ void k_noret(unsigned *ctr, int m) {
  for (int i = 0; i < m; ++i) if (threadIdx.x == 0) atomicAdd(ctr, 1u);
}
__global__ void k_ret_spread(unsigned *ctr, unsigned *sink, int m) {  // one counter per workgroup, 256 B apart
  unsigned acc = 0;
  for (int i = 0; i < m; ++i) { unsigned v = 0; if (threadIdx.x == 0) v = atomicAdd(ctr + (blockIdx.x % 1024) * 64, 1u); acc += __shfl(v, 0, 64); }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
int main() {
  unsigned *ctr, *sink; hipMalloc(&ctr, 1 << 20); hipMalloc(&sink, 64); hipMemset(ctr, 0, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves : {256, 2048, 8192}) for (int m : {16, 128}) {
    float ms[3];
    for (int v = 0; v < 3; ++v) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (v == 0) hipLaunchKernelGGL(k_ret, dim3(waves), dim3(64), 0, 0, ctr, sink, m);
        if (v == 1) hipLaunchKernelGGL(k_noret, dim3(waves), dim3(64), 0, 0, ctr, m);
        if (v == 2) hipLaunchKernelGGL(k_ret_spread, dim3(waves), dim3(64), 0, 0, ctr, sink, m);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[v], e0, e1);
      }
    }
    const double n = (double)waves * m;
    printf("waves %5d x %3d atomics: same-address returning %.1f ns/op, non-returning %.1f ns/op, 1024 addresses returning %.1f ns/op\n",
           waves, m, ms[0] * 1e6 / n, ms[1] * 1e6 / n, ms[2] * 1e6 / n);
  }
  return 0;
}
