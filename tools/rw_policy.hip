// Cache-policy sweep of a bare in-place read+write tile pass (gfx950): every combination of the load and the
// store instruction's sc0 / sc1 / nt bits on the two tile shapes of the bench headline's read+write pass.
// Question: a mixed pass tops out near 5.9 TB/s where a write-only pass streams at 7.0 -- is that
// the cache policy of the accesses?
//   hipcc -O3 --offload-arch=gfx950 tools/rw_policy.hip -o /tmp/rw_policy && /tmp/rw_policy 24 32 20 "4,5,6,7,8,9,10,11,12" "11,12,13,14,15,16,17,18,19" "6,13,14,15,16,17,18,19,20"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float vf4 __attribute__((ext_vector_type(4)));

struct Args { char *st; int n, q; int pos[9]; int outer[32]; int n_outer; };

template <int P> __device__ __forceinline__ vf4 ld(const char *p) {
  vf4 r;
  if constexpr (P == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r) : "v"(p) : "memory");
  else if constexpr (P == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(r) : "v"(p) : "memory");
  else if constexpr (P == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(r) : "v"(p) : "memory");
  else if constexpr (P == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(r) : "v"(p) : "memory");
  else if constexpr (P == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=&v"(r) : "v"(p) : "memory");
  else if constexpr (P == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=&v"(r) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=&v"(r) : "v"(p) : "memory");
  return r;
}
template <int P> __device__ __forceinline__ void st(char *p, vf4 v) {
  if constexpr (P == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  else if constexpr (P == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
  else if constexpr (P == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  else if constexpr (P == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
  else if constexpr (P == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
  else if constexpr (P == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}
static const char *kName[7] = {"-", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt", "sc0"};

template <int LP, int SP>
__global__ void __launch_bounds__(512) k_rw(const Args a) {
  const uint32_t tid = threadIdx.x;
  uint64_t off[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const uint32_t f = tid + 512u * u;  // float4 index inside the tile: 3 bits of row, 9 tile positions
    uint64_t o = (uint64_t)(f & 7u) << 4;
#pragma unroll
    for (int k = 0; k < 9; ++k) o |= (uint64_t)((f >> (3 + k)) & 1u) << (a.pos[k] + 3);
    off[u] = o;
  }
  char *sb = a.st + (((size_t)blockIdx.y << a.n) << 3);
  const uint32_t n_it = 1u << a.q, t0 = blockIdx.x << a.q;
  for (uint32_t it = 0; it < n_it; ++it) {
    const uint32_t t = t0 + it;
    uint64_t base = 0;
    for (int k = 0; k < a.n_outer; ++k) base |= (uint64_t)((t >> k) & 1u) << (a.outer[k] + 3);
    vf4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld<LP>(sb + base + off[u]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) { asm volatile("" : "+v"(v[u])); v[u].x += 1.0f; }
#pragma unroll
    for (int u = 0; u < 8; ++u) st<SP>(sb + base + off[u], v[u]);
  }
}

template <int LP, int SP>
static float run(const Args &a, int states, int reps, hipEvent_t e0, hipEvent_t e1) {
  const dim3 grid((1u << (a.n - 13)) >> a.q, states);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_rw<LP, SP>), grid, dim3(512), 0, 0, a);
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_rw<LP, SP>), grid, dim3(512), 0, 0, a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}
template <int LP, int... SPs>
static void row(const Args &a, int states, int reps, hipEvent_t e0, hipEvent_t e1) {
  printf("  load %-11s|", kName[LP]);
  (printf(" %6.2f", run<LP, SPs>(a, states, reps, e0, e1) * 1e3 / states), ...);
  printf("\n"); fflush(stdout);
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 24, states = argc > 2 ? atoi(argv[2]) : 32, reps = argc > 3 ? atoi(argv[3]) : 20;
  char *d; CK(hipMalloc(&d, ((size_t)states << n) * 8)); CK(hipMemset(d, 0, ((size_t)states << n) * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int ai = 4; ai < argc; ++ai) {
    Args a; a.st = d; a.n = n; a.q = 2;
    int k = 0; const char *c = argv[ai];
    for (; *c && k < 9;) { a.pos[k++] = atoi(c); while (*c && *c != ',') ++c; if (*c == ',') ++c; }
    if (k != 9) { printf("need 9 positions: %s\n", argv[ai]); continue; }
    uint32_t used = 15u; bool ok = true;
    for (int i = 0; i < 9; ++i) { if (a.pos[i] < 4 || a.pos[i] >= n || (used >> a.pos[i] & 1u)) ok = false; used |= 1u << a.pos[i]; }
    if (!ok) { printf("bad positions: %s\n", argv[ai]); continue; }
    a.n_outer = 0;
    for (int b = 4; b < n; ++b) if (!(used >> b & 1u)) a.outer[a.n_outer++] = b;
    printf("tile {0-3,%s}, n = %d, %d states per launch: us per state (%.1f us = 8 TB/s); columns = store policy: -, nt, sc1, sc0 sc1, sc1 nt, sc0 sc1 nt, sc0\n",
           argv[ai], n, states, 16.0 * std::ldexp(1.0, n) / 8e12 * 1e6);
    row<0, 0, 1, 2, 3, 4, 5, 6>(a, states, reps, e0, e1);
    row<1, 0, 1, 2, 3, 4, 5, 6>(a, states, reps, e0, e1);
    row<2, 0, 1, 2, 3, 4, 5, 6>(a, states, reps, e0, e1);
    row<3, 0, 1, 2, 3, 4, 5, 6>(a, states, reps, e0, e1);
    row<4, 0, 1, 2, 3, 4, 5, 6>(a, states, reps, e0, e1);
    row<5, 0, 1, 2, 3, 4, 5, 6>(a, states, reps, e0, e1);
    row<6, 0, 1, 2, 3, 4, 5, 6>(a, states, reps, e0, e1);
  }
  return 0;
}
