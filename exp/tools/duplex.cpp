// PCIe: H2D and D2H alone and together (pinned host memory, two streams): does this box move both directions at once?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
  const size_t n = 128u << 20;
  void *h1, *h2, *d1, *d2; hipHostMalloc(&h1, n, 0); hipHostMalloc(&h2, n, 0); hipMalloc(&d1, n); hipMalloc(&d2, n);
  hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  auto run = [&](bool up, bool down, int chunks) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 8; ++r) for (int c = 0; c < chunks; ++c) {
      const size_t o = c * (n / chunks);
      if (up) hipMemcpyAsync((char *)d1 + o, (char *)h1 + o, n / chunks, hipMemcpyHostToDevice, a);
      if (down) hipMemcpyAsync((char *)h2 + o, (char *)d2 + o, n / chunks, hipMemcpyDeviceToHost, b);
    }
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 8;
  };
  for (int chunks : {1, 64}) {
    run(true, true, chunks);
    const double u = run(true, false, chunks), dn = run(false, true, chunks), both = run(true, true, chunks);
    printf("chunks %2d: H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)  both %.2f ms (%.1f GB/s total)\n", chunks, u, n / u / 1e6, dn, n / dn / 1e6, both, 2.0 * n / both / 1e6);
  }
  return 0;
}
