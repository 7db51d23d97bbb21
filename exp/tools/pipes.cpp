// Which pairs of HIP streams run kernels concurrently?  N streams, a one-workgroup spin kernel of ~T µs on each of a pair at the same
// time: elapsed ≈ T when the two overlap, ≈ 2T when they share a hardware queue / pipe that serialises them.
// build: hipcc -O2 --offload-arch=gfx950 exp/tools/pipes.cpp -o exp/tools/pipes
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(long long cycles, int *sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 1024) *sink = 1;
}
int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 8, G = argc > 2 ? atoi(argv[2]) : 1;
  std::vector<hipStream_t> st(N);
  for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const long long cyc = 100 * 300;   // wall_clock64 ticks at 100 MHz: 300 µs
  for (auto &s : st) hipLaunchKernelGGL(spin, dim3(G), dim3(64), 0, s, 1000, nullptr);
  hipDeviceSynchronize();
  auto run = [&](std::vector<int> ids) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i : ids) hipLaunchKernelGGL(spin, dim3(G), dim3(64), 0, st[i], cyc, nullptr);
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  };
  printf("single: %.0f us\n", run({0}));
  printf("pair matrix (us):\n");
  for (int i = 0; i < N; ++i) {
    for (int j = 0; j < N; ++j) printf("%5.0f ", i == j ? 0.0 : run({i, j}));
    printf("\n");
  }
  std::vector<int> all; for (int i = 0; i < N; ++i) { all.push_back(i); printf("first %d together: %.0f us\n", i + 1, run(all)); }
  return 0;
}
