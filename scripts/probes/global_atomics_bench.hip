// micro-benchmark: global atomic add rates on gfx950, lane-distinct addresses spread over a film-sized buffer
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int MODE>  // MODE 0: atomicAdd, 1: plain store
__global__ __launch_bounds__(64) void k(T *buf, size_t n, int m) {
  const size_t wave = blockIdx.x;
  for (int i = 0; i < m; ++i) {
    // 27 planes of 16 pixels, like a G-BRE write-out: address = (pixel * 27 + k)
    const size_t pix = ((wave * 7919u + (size_t)i * 104729u) % (n / 27 / 16)) * 16;
    for (int idx = threadIdx.x; idx < 27 * 16; idx += 64) {
      const size_t a = (pix + idx % 16) * 27 + idx / 16;
      if (MODE == 0) atomicAdd(&buf[a], T(1));
      else buf[a] = T(1);
    }
  }
}
template <typename T, int MODE> float run(T *buf, size_t n, int m) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<T, MODE>), dim3(4096), dim3(64), 0, 0, buf, n, m);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}
int main() {
  const size_t n = (size_t)512 * 512 * 27;
  void *buf; (void)hipMalloc(&buf, n * 8); (void)hipMemset(buf, 0, n * 8);
  const int m = 64;
  const double ops = 4096.0 * m * 27 * 16;
  auto rep = [&](const char *name, float ms) { printf("%-28s %8.3f ms  -> %7.2f G ops/s\n", name, ms, ops / (ms * 1e-3) / 1e9); };
  rep("float atomicAdd", run<float, 0>((float *)buf, n, m));
  rep("double atomicAdd", run<double, 0>((double *)buf, n, m));
  rep("uint atomicAdd", run<unsigned, 0>((unsigned *)buf, n, m));
  rep("float plain store", run<float, 1>((float *)buf, n, m));
  return 0;
}
