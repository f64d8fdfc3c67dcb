// micro-benchmark: LDS atomic add rates on gfx950 (one wave per workgroup, 8 workgroups per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int MODE>  // MODE 0: lane-distinct addresses, 1: 8 lanes per address, 2: plain read-add-write, 3: returning
__global__ __launch_bounds__(64) void k(T *out, int m) {
  __shared__ T acc[27][64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 27 * 64; i += 64) (&acc[0][0])[i] = T(0);
  __syncthreads();
  const int b = MODE == 1 ? lane / 8 : lane;
  T v = T(lane + 1);
  for (int i = 0; i < m; ++i) {
#pragma unroll
    for (int kk = 0; kk < 27; ++kk) {
      if (MODE == 2) acc[kk][b] += v;
      else if (MODE == 3) v += atomicAdd(&acc[kk][lane], v) * T(1e-30);
      else atomicAdd(&acc[kk][b], v);
    }
    v += T(1);
  }
  __syncthreads();
  T s = T(0);
  for (int kk = 0; kk < 27; ++kk) s += acc[kk][lane];
  if (s == T(12345)) out[0] = s;
}
template <typename T, int MODE> float run(T *out, int m) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<T, MODE>), dim3(256 * 8), dim3(64), 0, 0, out, m);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}
int main() {
  void *out; (void)hipMalloc(&out, 64);
  const int m = 2000;
  // per CU: 8 waves x m x 27 instructions x 64 lanes
  const double lanesPerCU = 8.0 * m * 27 * 64;
  auto rep = [&](const char *name, float ms) { printf("%-44s %8.3f ms  -> %6.2f lanes/clk/CU (2.4 GHz)\n", name, ms, lanesPerCU / (ms * 1e-3 * 2.4e9)); };
  rep("float atomicAdd, lane-distinct addresses", run<float, 0>((float *)out, m));
  rep("float atomicAdd, 8 lanes per address", run<float, 1>((float *)out, m));
  rep("float plain read-add-write", run<float, 2>((float *)out, m));
  rep("uint atomicAdd, lane-distinct addresses", run<unsigned, 0>((unsigned *)out, m));
  rep("uint atomicAdd, 8 lanes per address", run<unsigned, 1>((unsigned *)out, m));
  rep("u64 atomicAdd, lane-distinct addresses", run<unsigned long long, 0>((unsigned long long *)out, m));
  rep("double atomicAdd, lane-distinct addresses", run<double, 0>((double *)out, m));
  rep("double atomicAdd, 8 lanes per address", run<double, 1>((double *)out, m));
  rep("float atomicAdd returning, lane-distinct", run<float, 3>((float *)out, m));
  rep("double atomicAdd returning, lane-distinct", run<double, 3>((double *)out, m));
  return 0;
}
