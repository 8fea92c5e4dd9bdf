// Does v_pk_fma_f32 double fp32 throughput on gfx950?  Tight dependent-free loops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k_scalar(float* out, int iters) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  const float m = 0.999f, c = 0.001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = fmaf(a[i], m, c);
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_packed(float* out, int iters) {
  f2 a[8];
  for (int i = 0; i < 8; ++i) a[i] = (f2){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  const f2 m = {0.999f, 0.998f}, c = {0.001f, 0.002f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* d; hipMalloc(&d, 256 * 2048 * 4 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int waves = 1; waves <= 4; waves *= 2) {
    dim3 grid(256 * 4), block(64 * waves);  // 4 blocks per CU
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0); hipLaunchKernelGGL(k_scalar, grid, block, 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = 2.0 * 16 * iters * grid.x * block.x;
      if (rep) printf("scalar fma  block=%4d: %7.3f ms %7.1f TFLOP/s\n", block.x, ms, flops / ms / 1e9);
      hipEventRecord(e0); hipLaunchKernelGGL(k_packed, grid, block, 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("packed fma  block=%4d: %7.3f ms %7.1f TFLOP/s\n", block.x, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
