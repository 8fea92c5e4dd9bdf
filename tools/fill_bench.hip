// Zero-fill rate of 4 GiB: one float4 per thread, four per thread (strided by 256), grid-stride
// loop, non-temporal variants, hipMemsetAsync.
//   hipcc -O3 --offload-arch=gfx950 tools/fill_bench.hip -o tools/fill_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ inline void st(float4 *p, float4 v) {
  if (NT) __builtin_nontemporal_store((f4){v.x, v.y, v.z, v.w}, reinterpret_cast<f4 *>(p));
  else *p = v;
}
template <bool NT> __global__ void __launch_bounds__(256) k_one(float4 *p) {
  st<NT>(p + (size_t)blockIdx.x * 256 + threadIdx.x, make_float4(0.f, 0.f, 0.f, 0.f));
}
template <bool NT> __global__ void __launch_bounds__(256) k_four(float4 *p) {
  float4 *q = p + (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) st<NT>(q + u * 256, make_float4(0.f, 0.f, 0.f, 0.f));
}
template <bool NT> __global__ void __launch_bounds__(256) k_sixteen(float4 *p) {
  float4 *q = p + (size_t)blockIdx.x * 4096 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 16; ++u) st<NT>(q + u * 256, make_float4(0.f, 0.f, 0.f, 0.f));
}
template <bool NT> __global__ void __launch_bounds__(256) k_loop(float4 *p, uint64_t count) {
  const uint64_t stride = (uint64_t)gridDim.x * 256u;
  for (uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x; k < count; k += stride)
    st<NT>(p + k, make_float4(0.f, 0.f, 0.f, 0.f));
}
int main() {
  const uint64_t count = 1ull << 28;  // float4 -> 4 GiB
  float4 *d;
  if (hipMalloc(&d, count * 16) != hipSuccess) return 1;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](const char *name, auto launch) {
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-34s %7.3f ms  %5.2f TB/s\n", name, best, count * 16.0 / best / 1e9);
  };
  run("one float4 per thread", [&] { hipLaunchKernelGGL(k_one<false>, dim3(count / 256), dim3(256), 0, 0, d); });
  run("one float4 per thread, nt", [&] { hipLaunchKernelGGL(k_one<true>, dim3(count / 256), dim3(256), 0, 0, d); });
  run("four per thread", [&] { hipLaunchKernelGGL(k_four<false>, dim3(count / 1024), dim3(256), 0, 0, d); });
  run("four per thread, nt", [&] { hipLaunchKernelGGL(k_four<true>, dim3(count / 1024), dim3(256), 0, 0, d); });
  run("sixteen per thread", [&] { hipLaunchKernelGGL(k_sixteen<false>, dim3(count / 4096), dim3(256), 0, 0, d); });
  run("sixteen per thread, nt", [&] { hipLaunchKernelGGL(k_sixteen<true>, dim3(count / 4096), dim3(256), 0, 0, d); });
  run("grid-stride loop, 8192 workgroups", [&] { hipLaunchKernelGGL(k_loop<false>, dim3(8192), dim3(256), 0, 0, d, count); });
  run("grid-stride loop, 2048 workgroups", [&] { hipLaunchKernelGGL(k_loop<false>, dim3(2048), dim3(256), 0, 0, d, count); });
  run("grid-stride loop nt, 8192", [&] { hipLaunchKernelGGL(k_loop<true>, dim3(8192), dim3(256), 0, 0, d, count); });
  run("hipMemsetAsync", [&] { (void)hipMemsetAsync(d, 0, count * 16, 0); });
  return 0;
}
