// Scratch probe (round 3): how much does a 1 GB device-to-host copy on a second stream slow the
// kernels of the first stream down, and by which copy path?  Prints kernel time alone / beside
// the copy, for hipMemcpyAsync (whatever path the runtime takes), for an SDMA copy issued
// through HSA directly, and for a small kernel that stores to mapped host memory.
//   hipcc -O3 --offload-arch=gfx950 d2h_interference.cpp -o d2h_interference -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define HK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { std::printf("%s: hsa status %d\n", #x, (int)s_); std::exit(1); } } while (0)

__global__ void k_stream(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = a[i]; v.x += 1; b[i] = v;
  }
}
__global__ void k_to_host(const uint4* __restrict__ a, uint4* __restrict__ h, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) h[i] = a[i];
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const size_t bytes = 1ull << 30, n = bytes / 16;
  const int wg_host = argc > 1 ? std::atoi(argv[1]) : 64;
  void *a, *b, *src, *host;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&src, bytes));
  CK(hipHostMalloc(&host, bytes, hipHostMallocDefault));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(src, 2, bytes));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, c0, c1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
  auto kernels = [&](int reps) { for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, s1, (const uint4*)a, (uint4*)b, n); };
  // warm
  kernels(2); CK(hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, s2)); CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventRecord(e0, s1)); kernels(40); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
  CK(hipEventElapsedTime(&ms, e0, e1)); std::printf("kernels alone: %.3f ms each (%.0f GB/s)\n", ms / 40, 2 * bytes / (ms / 40) / 1e6);
  const float alone = ms / 40;
  // (1) hipMemcpyAsync beside the kernels
  CK(hipEventRecord(c0, s2)); CK(hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, s2)); CK(hipEventRecord(c1, s2));
  CK(hipEventRecord(e0, s1)); kernels(40); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
  float cms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&cms, c0, c1));
  std::printf("hipMemcpyAsync D2H: copy %.2f ms (%.1f GB/s); 40 kernels %.2f ms = %.2f ms more than alone\n", cms, bytes / cms / 1e6, ms, ms - 40 * alone);
  // (1b) the same copy behind an event of the kernel stream (as the product issues it)
  {
    hipEvent_t dep; CK(hipEventCreateWithFlags(&dep, hipEventDisableTiming));
    kernels(1); CK(hipEventRecord(dep, s1)); CK(hipStreamWaitEvent(s2, dep, 0));
    CK(hipEventRecord(c0, s2)); CK(hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, s2)); CK(hipEventRecord(c1, s2));
    CK(hipEventRecord(e0, s1)); kernels(40); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&cms, c0, c1));
    std::printf("hipMemcpyAsync D2H behind hipStreamWaitEvent: copy %.2f ms (%.1f GB/s); 40 kernels %.2f ms = %.2f ms more than alone\n", cms, bytes / cms / 1e6, ms, ms - 40 * alone);
    // (1c) behind a blocking-sync event recorded after it, as ev_codes
    hipEvent_t done; CK(hipEventCreateWithFlags(&done, hipEventBlockingSync | hipEventDisableTiming));
    kernels(1); CK(hipEventRecord(dep, s1)); CK(hipStreamWaitEvent(s2, dep, 0));
    CK(hipEventRecord(c0, s2)); CK(hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, s2)); CK(hipEventRecord(c1, s2)); CK(hipEventRecord(done, s2));
    CK(hipEventRecord(e0, s1)); kernels(40); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(done)); CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&cms, c0, c1));
    std::printf("  ... and a blocking event after it: copy %.2f ms (%.1f GB/s); 40 kernels %.2f ms = %.2f ms more than alone\n", cms, bytes / cms / 1e6, ms, ms - 40 * alone);
    // (1d) two copies back to back (small one first)
    kernels(1); CK(hipEventRecord(dep, s1)); CK(hipStreamWaitEvent(s2, dep, 0));
    CK(hipMemcpyAsync(host, src, 128 << 20, hipMemcpyDeviceToHost, s2));
    CK(hipEventRecord(c0, s2)); CK(hipMemcpyAsync((char*)host + (128 << 20), (char*)src + (128 << 20), bytes - (128 << 20), hipMemcpyDeviceToHost, s2)); CK(hipEventRecord(c1, s2));
    CK(hipEventRecord(e0, s1)); kernels(40); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&cms, c0, c1));
    std::printf("  ... two copies back to back: second copy %.2f ms; 40 kernels %.2f ms = %.2f ms more than alone\n", cms, ms, ms - 40 * alone);
  }
  // (1e) the dependency NOT yet satisfied when the copy is submitted (40 kernels queued before the event)
  {
    hipEvent_t dep; CK(hipEventCreateWithFlags(&dep, hipEventDisableTiming));
    kernels(40); CK(hipEventRecord(dep, s1)); CK(hipStreamWaitEvent(s2, dep, 0));
    CK(hipEventRecord(c0, s2)); CK(hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, s2)); CK(hipEventRecord(c1, s2));
    CK(hipEventRecord(e0, s1)); kernels(40); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&cms, c0, c1));
    std::printf("hipMemcpyAsync D2H behind a PENDING event: copy %.2f ms (%.1f GB/s); 40 kernels %.2f ms = %.2f ms more than alone\n", cms, bytes / cms / 1e6, ms, ms - 40 * alone);
  }
  // (1g) the big copy followed by a 16-byte copy on the same stream (as the product's {state, flags} tail)
  {
    CK(hipEventRecord(c0, s2)); CK(hipMemcpyAsync(host, src, bytes - 4096, hipMemcpyDeviceToHost, s2));
    CK(hipMemcpyAsync((char*)host + bytes - 64, (char*)src + bytes - 64, 16, hipMemcpyDeviceToHost, s2)); CK(hipEventRecord(c1, s2));
    CK(hipEventRecord(e0, s1)); kernels(40); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&cms, c0, c1));
    std::printf("big D2H + 16-byte D2H behind it: copies %.2f ms; 40 kernels %.2f ms = %.2f ms more than alone\n", cms, ms, ms - 40 * alone);
  }
  // (1f) a size that is not a multiple of four bytes
  {
    const size_t odd = 1052146922ull < bytes ? 1052146922ull : bytes - 2;
    CK(hipEventRecord(c0, s2)); CK(hipMemcpyAsync(host, src, odd, hipMemcpyDeviceToHost, s2)); CK(hipEventRecord(c1, s2));
    CK(hipEventRecord(e0, s1)); kernels(40); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&cms, c0, c1));
    std::printf("hipMemcpyAsync D2H of %zu bytes (2 mod 4): copy %.2f ms (%.1f GB/s); 40 kernels %.2f ms = %.2f ms more than alone\n", odd, cms, odd / cms / 1e6, ms, ms - 40 * alone);
  }
  // (2) SDMA through HSA
  {
    hsa_amd_pointer_info_t pi_src, pi_dst; pi_src.size = sizeof pi_src; pi_dst.size = sizeof pi_dst;
    HK(hsa_amd_pointer_info(src, &pi_src, nullptr, nullptr, nullptr));
    HK(hsa_amd_pointer_info(host, &pi_dst, nullptr, nullptr, nullptr));
    hsa_signal_t sig; HK(hsa_signal_create(1, 0, nullptr, &sig));
    for (int force = 0; force < 2; ++force) {
      hsa_signal_store_relaxed(sig, 1);
      CK(hipEventRecord(e0, s1)); kernels(40); CK(hipEventRecord(e1, s1));
      const double t0 = now();
      hsa_status_t st;
      if (force) {
        uint32_t mask = 0;
        hsa_amd_memory_copy_engine_status(pi_dst.agentOwner, pi_src.agentOwner, &mask);
        std::printf("  free SDMA engines mask 0x%x\n", mask);
        st = hsa_amd_memory_async_copy_on_engine(host, pi_dst.agentOwner, src, pi_src.agentOwner, bytes, 0, nullptr, sig,
                                                 (hsa_amd_sdma_engine_id_t)(mask & (~mask + 1)), true);
      } else {
        st = hsa_amd_memory_async_copy(host, pi_dst.agentOwner, src, pi_src.agentOwner, bytes, 0, nullptr, sig);
      }
      if (st != HSA_STATUS_SUCCESS) { std::printf("hsa copy (force=%d) failed: %d\n", force, (int)st); CK(hipDeviceSynchronize()); continue; }
      while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) >= 1) {}
      const double t1 = now();
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::printf("HSA async copy (on_engine=%d): copy %.2f ms (%.1f GB/s); 40 kernels %.2f ms = %.2f ms more than alone\n", force, t1 - t0,
                  bytes / (t1 - t0) / 1e6, ms, ms - 40 * alone);
    }
  }
  // (3) a kernel that stores to the mapped host buffer, few workgroups
  for (int wg : {16, 64, 256, 1024}) {
    (void)wg_host;
    CK(hipEventRecord(c0, s2)); hipLaunchKernelGGL(k_to_host, dim3(wg), dim3(256), 0, s2, (const uint4*)src, (uint4*)host, n); CK(hipEventRecord(c1, s2));
    CK(hipEventRecord(e0, s1)); kernels(40); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&cms, c0, c1));
    std::printf("store-to-host kernel, %4d workgroups: copy %.2f ms (%.1f GB/s); 40 kernels %.2f ms = %.2f ms more than alone\n", wg, cms,
                bytes / cms / 1e6, ms, ms - 40 * alone);
  }
  return 0;
}
