// scratch: what a streaming copy reaches on this GPU, by grid size / items in flight / cache policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n16) {
  const uint64_t stride = (uint64_t)gridDim.x * 256u;
  uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) { typedef unsigned int v4 __attribute__((ext_vector_type(4))); const v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4*>(&src[i + u * stride])); v[u] = make_uint4(t.x, t.y, t.z, t.w); }
      else v[u] = src[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) { typedef unsigned int v4 __attribute__((ext_vector_type(4))); v4 t = {v[u].x, v[u].y, v[u].z, v[u].w}; __builtin_nontemporal_store(t, reinterpret_cast<v4*>(&dst[i + u * stride])); }
      else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}
// contiguous chunk per workgroup
template <int U>
__global__ __launch_bounds__(256) void k_copy_chunk(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n16) {
  const uint64_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const uint64_t b = (uint64_t)blockIdx.x * per, e = b + per < n16 ? b + per : n16;
  for (uint64_t i = b + threadIdx.x; i < e; i += 256u * U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256u < e) v[u] = src[i + u * 256u];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256u < e) dst[i + u * 256u] = v[u];
  }
}
int main() {
  const uint64_t bytes = 1ull << 30, n16 = bytes / 16;
  void *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.0f GB/s\n", name, 5 * 2.0 * bytes / (ms * 1e-3) / 1e9);
  };
  for (int g : {1024, 2048, 4096, 8192, 16384, 65536}) {
    char nm[64];
    snprintf(nm, 64, "stride U4 grid %d", g); run(nm, [&] { hipLaunchKernelGGL((k_copy<4, false>), dim3(g), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, n16); });
    snprintf(nm, 64, "stride U4 nt grid %d", g); run(nm, [&] { hipLaunchKernelGGL((k_copy<4, true>), dim3(g), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, n16); });
    snprintf(nm, 64, "stride U8 grid %d", g); run(nm, [&] { hipLaunchKernelGGL((k_copy<8, false>), dim3(g), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, n16); });
    snprintf(nm, 64, "chunk U4 grid %d", g); run(nm, [&] { hipLaunchKernelGGL((k_copy_chunk<4>), dim3(g), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, n16); });
  }
  { char nm[64]; snprintf(nm, 64, "one item per thread"); run(nm, [&] { hipLaunchKernelGGL((k_copy<1, false>), dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, n16); }); }
  run("hipMemcpyDtoD", [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
  return 0;
}
