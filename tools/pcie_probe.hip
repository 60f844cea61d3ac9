// Probe: how should a host-pointer call move 32 MB each way?  pageable hipMemcpy vs
// hipHostRegister + copy + unregister vs a persistent pinned staging buffer filled by memcpy.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t n = 32u << 20;
  char* host = (char*)aligned_alloc(4096, n);
  memset(host, 1, n);
  void* dev; hipMalloc(&dev, n);
  void* pinned; hipHostMalloc(&pinned, n, hipHostMallocDefault);
  hipStream_t st; hipStreamCreate(&st);
  for (int rep = 0; rep < 3; ++rep) {
    double t = now(); hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
    double a = now() - t;
    t = now(); hipMemcpyAsync(host, dev, n, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
    double b = now() - t;
    t = now(); hipHostRegister(host, n, hipHostRegisterDefault); double r1 = now() - t;
    t = now(); hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); double c = now() - t;
    t = now(); hipMemcpyAsync(host, dev, n, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); double d = now() - t;
    t = now(); hipHostUnregister(host); double r2 = now() - t;
    t = now(); memcpy(pinned, host, n); double m1 = now() - t;
    t = now();
    { std::vector<std::thread> th; const int T = 4; for (int i = 0; i < T; ++i) th.emplace_back([&, i] { memcpy((char*)pinned + n / T * i, host + n / T * i, n / T); }); for (auto& x : th) x.join(); }
    double m4 = now() - t;
    t = now(); hipMemcpyAsync(dev, pinned, n, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); double e = now() - t;
    printf("rep %d: pageable H2D %.2f ms D2H %.2f ms | register %.2f ms + H2D %.2f + D2H %.2f + unregister %.2f | memcpy->pinned 1 thr %.2f ms, 4 thr %.2f ms, pinned H2D %.2f ms\n",
           rep, a * 1e3, b * 1e3, r1 * 1e3, c * 1e3, d * 1e3, r2 * 1e3, m1 * 1e3, m4 * 1e3, e * 1e3);
  }
  return 0;
}
