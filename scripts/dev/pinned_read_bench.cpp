// Development: how fast does the CPU read page-locked host memory of each hipHostMalloc kind?
//   hipcc -O2 scripts/dev/pinned_read_bench.cpp -o /tmp/pinned_read_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static double sum_ms(const unsigned* p, size_t words, unsigned long long* out) {
  const auto t0 = std::chrono::steady_clock::now();
  unsigned long long s = 0;
  for (size_t i = 0; i < words; ++i) s += p[i];
  *out = s;
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
  const size_t bytes = 256u << 20, words = bytes / 4;
  void* d = nullptr;
  if (hipMalloc(&d, bytes) != hipSuccess) return 1;
  (void)hipMemset(d, 1, bytes);
  struct { const char* name; unsigned flags; } kinds[] = {
      {"hipHostMallocDefault", hipHostMallocDefault}, {"hipHostMallocNonCoherent", hipHostMallocNonCoherent},
      {"hipHostMallocCoherent", hipHostMallocCoherent}, {"hipHostMallocNumaUser|Default", hipHostMallocDefault}};
  for (int k = 0; k < 3; ++k) {
    void* h = nullptr;
    if (hipHostMalloc(&h, bytes, kinds[k].flags) != hipSuccess) { std::printf("%s: allocation failed\n", kinds[k].name); continue; }
    const auto t0 = std::chrono::steady_clock::now();
    (void)hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
    const double copy_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long s1, s2;
    const double a = sum_ms(static_cast<unsigned*>(h), words, &s1), b = sum_ms(static_cast<unsigned*>(h), words, &s2);
    std::printf("%-28s D2H %.1f ms (%.1f GB/s), CPU read 1st %.1f ms (%.1f GB/s), 2nd %.1f ms, sum %llu\n", kinds[k].name, copy_ms,
                bytes / copy_ms / 1e6, a, bytes / a / 1e6, b, s1 + s2);
    (void)hipHostFree(h);
  }
  void* m = std::malloc(bytes);
  std::memset(m, 1, bytes);
  unsigned long long s;
  const double a = sum_ms(static_cast<unsigned*>(m), words, &s);
  std::printf("%-28s CPU read %.1f ms (%.1f GB/s)\n", "malloc", a, bytes / a / 1e6);
  return 0;
}
