// Issue rate of the dense-gate asm block of k_tile2 (64 v_pk_mul/fma_f32 per 16 amplitudes) in
// isolation: matrix entries as SGPR pairs (the product's form) vs VGPR pairs, with and without
// the op_sel / neg modifiers, 4 vs 8 interleaved dependency chains.  Registers only, no memory.
//   hipcc -O3 --offload-arch=gfx950 tools/gate_issue_bench.hip -o tools/gate_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;

#define PAIR2(a0, a1, a2, a3, M0, M1, M2, M3, C)                                                   \
  asm volatile(                                                                                    \
      "v_pk_mul_f32 %4, %8, %0 op_sel_hi:[0,1]\n\t"                                                \
      "v_pk_mul_f32 %5, %10, %0 op_sel_hi:[0,1]\n\t"                                               \
      "v_pk_mul_f32 %6, %8, %2 op_sel_hi:[0,1]\n\t"                                                \
      "v_pk_mul_f32 %7, %10, %2 op_sel_hi:[0,1]\n\t"                                               \
      "v_pk_fma_f32 %4, %8, %0, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"            \
      "v_pk_fma_f32 %5, %10, %0, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"           \
      "v_pk_fma_f32 %6, %8, %2, %6 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"            \
      "v_pk_fma_f32 %7, %10, %2, %7 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"           \
      "v_pk_fma_f32 %4, %9, %1, %4 op_sel_hi:[0,1,1]\n\t"                                          \
      "v_pk_fma_f32 %5, %11, %1, %5 op_sel_hi:[0,1,1]\n\t"                                         \
      "v_pk_fma_f32 %6, %9, %3, %6 op_sel_hi:[0,1,1]\n\t"                                          \
      "v_pk_fma_f32 %7, %11, %3, %7 op_sel_hi:[0,1,1]\n\t"                                         \
      "v_pk_fma_f32 %0, %9, %1, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"            \
      "v_pk_fma_f32 %1, %11, %1, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"           \
      "v_pk_fma_f32 %2, %9, %3, %6 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"            \
      "v_pk_fma_f32 %3, %11, %3, %7 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"           \
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)         \
      : C(M0), C(M1), C(M2), C(M3))
#define PLAIN2(a0, a1, a2, a3, M0, M1, M2, M3, C)                                                  \
  asm volatile(                                                                                    \
      "v_pk_mul_f32 %4, %8, %0\n\tv_pk_mul_f32 %5, %10, %0\n\t"                                    \
      "v_pk_mul_f32 %6, %8, %2\n\tv_pk_mul_f32 %7, %10, %2\n\t"                                    \
      "v_pk_fma_f32 %4, %8, %0, %4\n\tv_pk_fma_f32 %5, %10, %0, %5\n\t"                            \
      "v_pk_fma_f32 %6, %8, %2, %6\n\tv_pk_fma_f32 %7, %10, %2, %7\n\t"                            \
      "v_pk_fma_f32 %4, %9, %1, %4\n\tv_pk_fma_f32 %5, %11, %1, %5\n\t"                            \
      "v_pk_fma_f32 %6, %9, %3, %6\n\tv_pk_fma_f32 %7, %11, %3, %7\n\t"                            \
      "v_pk_fma_f32 %0, %9, %1, %4\n\tv_pk_fma_f32 %1, %11, %1, %5\n\t"                            \
      "v_pk_fma_f32 %2, %9, %3, %6\n\tv_pk_fma_f32 %3, %11, %3, %7\n\t"                            \
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)         \
      : C(M0), C(M1), C(M2), C(M3))
#define CS "s"
#define CV "v"

template <int MODE>  // 0 SGPR + modifiers (product), 1 VGPR + modifiers, 2 SGPR plain, 3 VGPR plain
__global__ void __launch_bounds__(256) k_gate(u64 *out, const u64 *mats, int iters) {
  u64 r[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) r[c] = (u64)(threadIdx.x + c) * 0x3f8000003f800000ull;
  u64 t0, t1, t2, t3;
  const u64 m0 = mats[0], m1 = mats[1], m2 = mats[2], m3 = mats[3];
  u64 v0 = m0 + threadIdx.x * 0, v1 = m1, v2 = m2, v3 = m3;
  asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (MODE == 0) PAIR2(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3], m0, m1, m2, m3, CS);
      if (MODE == 1) PAIR2(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3], v0, v1, v2, v3, CV);
      if (MODE == 2) PLAIN2(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3], m0, m1, m2, m3, CS);
      if (MODE == 3) PLAIN2(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3], v0, v1, v2, v3, CV);
    }
  }
  u64 s = 0;
#pragma unroll
  for (int c = 0; c < 16; ++c) s ^= r[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  u64 *d, *m;
  const int blocks = 256 * 5;
  hipMalloc(&d, (size_t)blocks * 256 * 8);
  hipMalloc(&m, 64);
  u64 hm[4] = {0x3f0000003f000000ull, 0x3e8000003e800000ull, 0x3e0000003f000000ull, 0x3d8000003e000000ull};
  hipMemcpy(m, hm, 32, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  const char *names[4] = {"SGPR matrix, op_sel/neg (product)", "VGPR matrix, op_sel/neg", "SGPR matrix, plain", "VGPR matrix, plain"};
  for (int mode = 0; mode < 4; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k_gate<0>, dim3(blocks), dim3(256), 0, 0, d, m, iters);
      if (mode == 1) hipLaunchKernelGGL(k_gate<1>, dim3(blocks), dim3(256), 0, 0, d, m, iters);
      if (mode == 2) hipLaunchKernelGGL(k_gate<2>, dim3(blocks), dim3(256), 0, 0, d, m, iters);
      if (mode == 3) hipLaunchKernelGGL(k_gate<3>, dim3(blocks), dim3(256), 0, 0, d, m, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double instr = 64.0 * iters * blocks * 4;  // wave-instructions
    const double flops = instr * 64 * 4;             // 64 lanes x 2 fma x 2
    printf("%-36s %8.3f ms  %6.1f TFLOP/s  %5.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", names[mode], best,
           flops / best / 1e9, best * 1e-3 * 2.4e9 / (instr / 1024.0));
  }
  return 0;
}
