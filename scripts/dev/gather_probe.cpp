// scratch: how many random reads of a 256 MiB array does the GPU serve per second?  (the finisher's first pass makes
// 231 M of them, 24..56 bytes each, in 12..14 ms)
// build on the GPU box: hipcc -O3 --offload-arch=gfx950 scripts/dev/gather_probe.cpp -o /tmp/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;
__device__ __forceinline__ u32 mix(u32 x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// W: aligned 8-byte words read per item; E items per thread, all loads issued before any is used
template <int W, int E>
__global__ __launch_bounds__(256) void k_gather(const u64* __restrict__ T, u32 nwords, u32 items, u64* __restrict__ out, u32 span_mask) {
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  u64 acc = 0;
  u64 v[E][W];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const u32 i = t + (u32)e * gridDim.x * 256u;
    const u32 w = (mix(i) & span_mask) % (nwords - W);
#pragma unroll
    for (int q = 0; q < W; ++q) v[e][q] = T[w + q];
  }
#pragma unroll
  for (int e = 0; e < E; ++e)
#pragma unroll
    for (int q = 0; q < W; ++q) acc ^= v[e][q];
  if (acc == 0x123456789ull) out[t] = acc;
}
int main() {
  const u64 bytes = 1ull << 28; const u32 nwords = (u32)(bytes / 8);
  u64 *T, *out; hipMalloc(&T, bytes); hipMalloc(&out, 1 << 20); hipMemset(T, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const u32 items = 231u << 20;
  auto run = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 3; ++r) launch(); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.2f ms  (%.1f G items/s)\n", name, ms / 3, items / (ms / 3 * 1e-3) / 1e9);
  };
  for (u32 mask : {0xFFFFFFFFu, (1u << 24) - 1u, (1u << 20) - 1u}) {
    printf("-- random words within %s\n", mask == 0xFFFFFFFFu ? "the whole 256 MiB" : mask == (1u << 24) - 1u ? "128 MiB" : "8 MiB");
    run("3 words, 1 item per thread", [&] { hipLaunchKernelGGL((k_gather<3, 1>), dim3(items / 256), dim3(256), 0, 0, T, nwords, items, out, mask); });
    run("3 words, 4 items per thread", [&] { hipLaunchKernelGGL((k_gather<3, 4>), dim3(items / 1024), dim3(256), 0, 0, T, nwords, items, out, mask); });
    run("1 word, 4 items per thread", [&] { hipLaunchKernelGGL((k_gather<1, 4>), dim3(items / 1024), dim3(256), 0, 0, T, nwords, items, out, mask); });
    run("6 words, 4 items per thread", [&] { hipLaunchKernelGGL((k_gather<6, 4>), dim3(items / 1024), dim3(256), 0, 0, T, nwords, items, out, mask); });
  }
  return 0;
}
