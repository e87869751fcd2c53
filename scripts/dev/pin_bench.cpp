// How long does it take to get 1.2 GB of page-locked host memory, and does the way matter for the
// device-to-host copy?  (a) hipHostMalloc, (b) 2 MiB-aligned malloc + MADV_HUGEPAGE + touch +
// hipHostRegister, (c) the same without the huge-page advice.  Build: hipcc -O2 pin_bench.cpp -o pin_bench
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static double ms_since(std::chrono::steady_clock::time_point t) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const size_t n = (size_t)1200 << 20;
  void* d = nullptr;
  CK(hipMalloc(&d, n));
  CK(hipMemset(d, 1, n));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (int rep = 0; rep < 3; ++rep) {
    auto t = std::chrono::steady_clock::now();
    void* h = nullptr;
    CK(hipHostMalloc(&h, n, hipHostMallocDefault));
    const double a = ms_since(t);
    t = std::chrono::steady_clock::now();
    CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    const double c = ms_since(t);
    t = std::chrono::steady_clock::now();
    CK(hipHostFree(h));
    std::printf("hipHostMalloc: alloc %.1f ms, copy %.1f ms (%.1f GB/s), free %.1f ms\n", a, c, n / c / 1e6, ms_since(t));
  }
  for (int huge = 1; huge >= 0; --huge)
    for (int rep = 0; rep < 3; ++rep) {
      auto t = std::chrono::steady_clock::now();
      void* h = nullptr;
      if (posix_memalign(&h, (size_t)2 << 20, n)) return 1;
      if (huge) madvise(h, n, MADV_HUGEPAGE);
      const double a0 = ms_since(t);
      t = std::chrono::steady_clock::now();
      for (size_t i = 0; i < n; i += 4096) static_cast<volatile char*>(h)[i] = 0;
      const double a1 = ms_since(t);
      t = std::chrono::steady_clock::now();
      CK(hipHostRegister(h, n, hipHostRegisterDefault));
      const double a2 = ms_since(t);
      t = std::chrono::steady_clock::now();
      CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      const double c = ms_since(t);
      t = std::chrono::steady_clock::now();
      CK(hipHostUnregister(h));
      std::free(h);
      std::printf("malloc%s + register: alloc %.1f ms, touch %.1f ms, register %.1f ms, copy %.1f ms (%.1f GB/s), release %.1f ms\n",
                  huge ? " + MADV_HUGEPAGE" : "", a0, a1, a2, c, n / c / 1e6, ms_since(t));
    }
  // registering WITHOUT touching first (the driver faults the pages in)
  for (int rep = 0; rep < 2; ++rep) {
    auto t = std::chrono::steady_clock::now();
    void* h = nullptr;
    if (posix_memalign(&h, (size_t)2 << 20, n)) return 1;
    madvise(h, n, MADV_HUGEPAGE);
    CK(hipHostRegister(h, n, hipHostRegisterDefault));
    const double a = ms_since(t);
    t = std::chrono::steady_clock::now();
    CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    const double c = ms_since(t);
    CK(hipHostUnregister(h));
    std::free(h);
    std::printf("malloc + MADV_HUGEPAGE + register untouched: %.1f ms, copy %.1f ms (%.1f GB/s)\n", a, c, n / c / 1e6);
  }
  return 0;
}
