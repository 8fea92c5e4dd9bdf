// Stand-alone experiments for the dense tile pass (round 2), n = 24, B states, in place:
// load a 2^12 tile -> LDS, G register-tile groups of four dense 2x2 gates each, store.
//   hipcc -O3 --offload-arch=gfx950 tools/tile2_bench.hip -o tools/tile2_bench
//   tools/tile2_bench [B=32] [L=5] [workgroups of the persistent variant=1280]     (-DQUAD: 4 pairs interleaved)
// The list `modes` in main() picks what runs; a mode is printed as `asm=<mode>`:
//    0  hipcc's version of the gate arithmetic (matrices from a uniform pointer)
//    1  packed-fp32 asm gate blocks, matrices in SGPRs (what k_tile2 uses)
//    2  the same in a persistent workgroup with the next tile prefetched in registers (k_t2p)
//    3  matrix core: the group's gates merged into a 32x32 real operator, v_mfma_f32_32x32x2_f32
//    5  mixed: (tile + group) % MIX_MOD < MIX_MFMA -> matrix core, else vector asm
//   11 / 14 / 15  = 1 / 3 / 5 without HBM traffic (compute only)
//   12  gather + scatter only (no gates);  13  gates only (no LDS traffic inside the groups)
//   21 / 22  = 1 / 2 as a read-only pass (no stores, one float per wave written)
// Every mode checks state 0 against a host reference (max|err|) except the compute-only ones,
// whose input is synthetic.  Results: profiles/r02_mfma_ab_tile2_bench.txt,
// r02_group_cost_split.txt, r02_mfma_mixed_concurrency.txt; DESIGN.md section 9b.
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;
typedef float vf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t sw(uint32_t e) { return e ^ (((e >> 5) & 15u) << 1); }
__device__ __forceinline__ uint32_t ins0(uint32_t i, int p) {
  return ((i >> p) << (p + 1)) | (i & ((1u << p) - 1u));
}

// two amplitude pairs (a0,a1), (a2,a3) under the same 2x2 matrix; T* are scratch pairs
#define PAIR2(a0, a1, a2, a3)                                                                    \
  asm volatile(                                                                                  \
      "v_pk_mul_f32 %4, %8, %0 op_sel_hi:[0,1]\n\t"                                              \
      "v_pk_mul_f32 %5, %10, %0 op_sel_hi:[0,1]\n\t"                                             \
      "v_pk_mul_f32 %6, %8, %2 op_sel_hi:[0,1]\n\t"                                              \
      "v_pk_mul_f32 %7, %10, %2 op_sel_hi:[0,1]\n\t"                                             \
      "v_pk_fma_f32 %4, %8, %0, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"          \
      "v_pk_fma_f32 %5, %10, %0, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"         \
      "v_pk_fma_f32 %6, %8, %2, %6 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"          \
      "v_pk_fma_f32 %7, %10, %2, %7 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"         \
      "v_pk_fma_f32 %4, %9, %1, %4 op_sel_hi:[0,1,1]\n\t"                                        \
      "v_pk_fma_f32 %5, %11, %1, %5 op_sel_hi:[0,1,1]\n\t"                                       \
      "v_pk_fma_f32 %6, %9, %3, %6 op_sel_hi:[0,1,1]\n\t"                                        \
      "v_pk_fma_f32 %7, %11, %3, %7 op_sel_hi:[0,1,1]\n\t"                                       \
      "v_pk_fma_f32 %0, %9, %1, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"          \
      "v_pk_fma_f32 %1, %11, %1, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"         \
      "v_pk_fma_f32 %2, %9, %3, %6 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"          \
      "v_pk_fma_f32 %3, %11, %3, %7 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"         \
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)       \
      : "s"(m00), "s"(m01), "s"(m10), "s"(m11))


#define PAIR4(a0, a1, a2, a3, a4, a5, a6, a7)                                                    \
  asm volatile(                                                                                  \
      "v_pk_mul_f32 %8, %16, %0 op_sel_hi:[0,1]\n\t"                                             \
      "v_pk_mul_f32 %9, %18, %0 op_sel_hi:[0,1]\n\t"                                             \
      "v_pk_mul_f32 %10, %16, %2 op_sel_hi:[0,1]\n\t"                                            \
      "v_pk_mul_f32 %11, %18, %2 op_sel_hi:[0,1]\n\t"                                            \
      "v_pk_mul_f32 %12, %16, %4 op_sel_hi:[0,1]\n\t"                                            \
      "v_pk_mul_f32 %13, %18, %4 op_sel_hi:[0,1]\n\t"                                            \
      "v_pk_mul_f32 %14, %16, %6 op_sel_hi:[0,1]\n\t"                                            \
      "v_pk_mul_f32 %15, %18, %6 op_sel_hi:[0,1]\n\t"                                            \
      "v_pk_fma_f32 %8, %16, %0, %8 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"         \
      "v_pk_fma_f32 %9, %18, %0, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"         \
      "v_pk_fma_f32 %10, %16, %2, %10 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"       \
      "v_pk_fma_f32 %11, %18, %2, %11 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"       \
      "v_pk_fma_f32 %12, %16, %4, %12 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"       \
      "v_pk_fma_f32 %13, %18, %4, %13 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"       \
      "v_pk_fma_f32 %14, %16, %6, %14 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"       \
      "v_pk_fma_f32 %15, %18, %6, %15 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"       \
      "v_pk_fma_f32 %8, %17, %1, %8 op_sel_hi:[0,1,1]\n\t"                                       \
      "v_pk_fma_f32 %9, %19, %1, %9 op_sel_hi:[0,1,1]\n\t"                                       \
      "v_pk_fma_f32 %10, %17, %3, %10 op_sel_hi:[0,1,1]\n\t"                                     \
      "v_pk_fma_f32 %11, %19, %3, %11 op_sel_hi:[0,1,1]\n\t"                                     \
      "v_pk_fma_f32 %12, %17, %5, %12 op_sel_hi:[0,1,1]\n\t"                                     \
      "v_pk_fma_f32 %13, %19, %5, %13 op_sel_hi:[0,1,1]\n\t"                                     \
      "v_pk_fma_f32 %14, %17, %7, %14 op_sel_hi:[0,1,1]\n\t"                                     \
      "v_pk_fma_f32 %15, %19, %7, %15 op_sel_hi:[0,1,1]\n\t"                                     \
      "v_pk_fma_f32 %0, %17, %1, %8 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"         \
      "v_pk_fma_f32 %1, %19, %1, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"         \
      "v_pk_fma_f32 %2, %17, %3, %10 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"        \
      "v_pk_fma_f32 %3, %19, %3, %11 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"        \
      "v_pk_fma_f32 %4, %17, %5, %12 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"        \
      "v_pk_fma_f32 %5, %19, %5, %13 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"        \
      "v_pk_fma_f32 %6, %17, %7, %14 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"        \
      "v_pk_fma_f32 %7, %19, %7, %15 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"        \
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),          \
        "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)   \
      : "s"(m00), "s"(m01), "s"(m10), "s"(m11))

template <int TB>
__device__ __forceinline__ void gate16(u64 (&a)[16], u64 m00, u64 m01, u64 m10, u64 m11) {
  u64 t0, t1, t2, t3;
  constexpr int S = 1 << TB;
  // the 8 pairs (c, c | S), two at a time
  int idx[8], k = 0;
#pragma unroll
  for (int c = 0; c < 16; ++c)
    if (!(c & S)) idx[k++] = c;
#ifdef QUAD
  u64 t4, t5, t6, t7;
#pragma unroll
  for (int q = 0; q < 8; q += 4)
    PAIR4(a[idx[q]], a[idx[q] | S], a[idx[q + 1]], a[idx[q + 1] | S], a[idx[q + 2]], a[idx[q + 2] | S],
          a[idx[q + 3]], a[idx[q + 3] | S]);
#else
#pragma unroll
  for (int q = 0; q < 8; q += 2) PAIR2(a[idx[q]], a[idx[q] | S], a[idx[q + 1]], a[idx[q + 1] | S]);
#endif
}

struct Args {
  float2 *states;
  const float *mats;  // [B][G][4][8]
  const float *amat;  // [B][G][16][64]: A-operand fragments of the merged 32x32 real matrix
  int n, T, L, G;
  int tile_bits[16];   // global positions of the local bits
  int outer_bits[32];
  int gbits[8][4];     // per group: 4 tile-local bits, ascending
  int use_asm;
  int mix_mod, mix_mfma;  // modes 5 / 15: (tile + group) % mix_mod < mix_mfma -> matrix-core group
  int ro;   // read-only pass: no stores, one float per workgroup and tile written instead
  float *sink;
};

template <bool ASM>
__global__ void __launch_bounds__(256) k_t2(const Args a) {
  extern __shared__ float4 smem4[];
  float2 *s = reinterpret_cast<float2 *>(smem4);
  const int tid = threadIdx.x, b = blockIdx.y;
  const uint32_t tile = blockIdx.x;
  const int T = a.T;
  uint64_t base = 0;
  for (int i = 0; i < a.n - T; ++i) base |= (uint64_t)((tile >> i) & 1u) << a.outer_bits[i];
  float2 *st = a.states + ((size_t)b << a.n) + base;
  // element pairs: j = 2 * (tid + u * 256), u = 0..7: bits 1..8 from tid, 9..11 from u
  const uint32_t jl = 2u * tid;  // local bits 0..8
  uint32_t goff = jl & ((1u << a.L) - 1u);
  for (int p = a.L; p < 9; ++p) goff |= ((jl >> p) & 1u) << a.tile_bits[p];
  uint32_t uoff[8];
#pragma unroll
  for (int u = 0; u < 8; ++u)
    uoff[u] = ((u & 1) << a.tile_bits[9]) | (((u >> 1) & 1) << a.tile_bits[10]) | (((u >> 2) & 1) << a.tile_bits[11]);
  const bool nomem = a.use_asm >= 10;
  {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (nomem) { v[u] = make_float4(1e-4f * tid, 1e-4f * u, 1e-4f, 2e-4f); continue; }
      const vf4 w = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(st + (goff | uoff[u])));
      v[u] = make_float4(w.x, w.y, w.z, w.w);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) reinterpret_cast<float4 *>(s)[sw(jl + u * 512u) >> 1] = v[u];
  }
  __syncthreads();
  const float *mrow = a.mats + (size_t)b * a.G * 32;
  for (int g = 0; g < a.G; ++g) {
    const int b0 = a.gbits[g][0], b1 = a.gbits[g][1], b2 = a.gbits[g][2], b3 = a.gbits[g][3];
    uint32_t off[16];
#pragma unroll
    for (int c = 0; c < 16; ++c)
      off[c] = sw(((c & 1) ? (1u << b0) : 0u) | ((c & 2) ? (1u << b1) : 0u) |
                  ((c & 4) ? (1u << b2) : 0u) | ((c & 8) ? (1u << b3) : 0u));
    const uint32_t bs = sw(ins0(ins0(ins0(ins0(tid, b0), b1), b2), b3));
    const u64 *m = reinterpret_cast<const u64 *>(mrow + g * 32);
    const bool mixed_mfma = (a.use_asm == 15 || a.use_asm == 5) && (((blockIdx.x + g) % a.mix_mod) < a.mix_mfma);
    if (ASM && (a.use_asm == 3 || a.use_asm == 14 || mixed_mfma)) {
      // matrix-core group: the 4 gates merged into one 16x16 complex operator = 32x32 real
      // matrix R; out[32 reals x 32 items] = R . in via 16 x v_mfma_f32_32x32x2_f32 per block of
      // 32 work items (two blocks per wave)
      typedef float f16v __attribute__((ext_vector_type(16)));
      const int lane = tid & 63, hh = lane >> 5;
      const float *A = a.amat + ((size_t)(b * a.G + g) * 16) * 64 + lane;
      float af[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) af[k] = A[k * 64];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const uint32_t item = (tid & ~63u) + 32u * blk + (lane & 31);
        const uint32_t bsi = sw(ins0(ins0(ins0(ins0(item, b0), b1), b2), b3));
        float bf[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) bf[k] = reinterpret_cast<const float *>(s)[2 * (bsi ^ off[k]) + hh];
        f16v acc = {0};
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k], bf[k], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
          const int a0 = 4 * (v >> 2) + ((v & 3) >> 1);  // lane half 0; half 1 holds amplitude a0 + 2
          s[bsi ^ (hh ? off[a0 + 2] : off[a0])] = make_float2(acc[v], acc[v + 1]);
        }
      }
    } else if (ASM) {
      u64 v[16];
      const bool lds_on = a.use_asm != 13 || g == 0, gates_on = a.use_asm != 12;
      if (lds_on) {
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = reinterpret_cast<const u64 *>(s)[bs ^ off[c]];
      } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = 0x3c0000003c000000ull + c + tid;
      }
      if (gates_on) {
        gate16<0>(v, m[0], m[1], m[2], m[3]);
        gate16<1>(v, m[4], m[5], m[6], m[7]);
        gate16<2>(v, m[8], m[9], m[10], m[11]);
        gate16<3>(v, m[12], m[13], m[14], m[15]);
      }
      if (lds_on || g + 1 == a.G) {
#pragma unroll
        for (int c = 0; c < 16; ++c) reinterpret_cast<u64 *>(s)[bs ^ off[c]] = v[c];
      } else {
        u64 x = 0;
#pragma unroll
        for (int c = 0; c < 16; ++c) x ^= v[c];
        if (x == 0x1234567ull) reinterpret_cast<u64 *>(s)[bs] = x;
      }
    } else {
      float2 v[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) v[c] = s[bs ^ off[c]];
      const float *mf = mrow + g * 32;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float2 m00 = make_float2(mf[t * 8 + 0], mf[t * 8 + 1]), m01 = make_float2(mf[t * 8 + 2], mf[t * 8 + 3]);
        const float2 m10 = make_float2(mf[t * 8 + 4], mf[t * 8 + 5]), m11 = make_float2(mf[t * 8 + 6], mf[t * 8 + 7]);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          if (c & (1 << t)) continue;
          const float2 x = v[c], y = v[c | (1 << t)];
          v[c] = make_float2(m00.x * x.x - m00.y * x.y + m01.x * y.x - m01.y * y.y,
                             m00.x * x.y + m00.y * x.x + m01.x * y.y + m01.y * y.x);
          v[c | (1 << t)] = make_float2(m10.x * x.x - m10.y * x.y + m11.x * y.x - m11.y * y.y,
                                        m10.x * x.y + m10.y * x.x + m11.x * y.y + m11.y * y.x);
        }
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) s[bs ^ off[c]] = v[c];
    }
    __syncthreads();
  }
  {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<float4 *>(s)[sw(jl + u * 512u) >> 1];
    if (a.ro) {
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
      if ((tid & 63) == 0) a.sink[((size_t)b * gridDim.x + blockIdx.x) * 4 + (tid >> 6)] = acc;
      return;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const vf4 w = {v[u].x, v[u].y, v[u].z, v[u].w};
      if (nomem && w.x != 12345.f) continue;
      __builtin_nontemporal_store(w, reinterpret_cast<vf4 *>(st + (goff | uoff[u])));
    }
  }
}


// Persistent variant: a workgroup walks tiles w, w + W, w + 2W, ...; the next tile's 8 float4 per
// lane are in flight (registers) while the gate groups run on the current tile in LDS.
__global__ void __launch_bounds__(256) k_t2p(const Args a, uint32_t n_work) {
  extern __shared__ float4 smem4[];
  float2 *s = reinterpret_cast<float2 *>(smem4);
  const int tid = threadIdx.x;
  const int T = a.T;
  const uint32_t tiles_per_state = 1u << (a.n - T);
  const uint32_t jl = 2u * tid;
  uint32_t goff = jl & ((1u << a.L) - 1u);
  for (int p = a.L; p < 9; ++p) goff |= ((jl >> p) & 1u) << a.tile_bits[p];
  uint32_t uoff[8];
#pragma unroll
  for (int u = 0; u < 8; ++u)
    uoff[u] = ((u & 1) << a.tile_bits[9]) | (((u >> 1) & 1) << a.tile_bits[10]) | (((u >> 2) & 1) << a.tile_bits[11]);
  auto tile_ptr = [&](uint32_t w) -> float2 * {
    const uint32_t tile = w & (tiles_per_state - 1u), b = w >> (a.n - T);
    uint64_t base = 0;
    for (int i = 0; i < a.n - T; ++i) base |= (uint64_t)((tile >> i) & 1u) << a.outer_bits[i];
    return a.states + ((size_t)b << a.n) + base + goff;
  };
  uint32_t w = blockIdx.x;
  if (w >= n_work) return;
  float racc = 0.f;
  float4 v[8];
  float2 *cur = tile_ptr(w);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const vf4 x = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(cur + uoff[u]));
    v[u] = make_float4(x.x, x.y, x.z, x.w);
  }
  for (; w < n_work; w += gridDim.x) {
    const int b = w >> (a.n - T);
#pragma unroll
    for (int u = 0; u < 8; ++u) reinterpret_cast<float4 *>(s)[sw(jl + u * 512u) >> 1] = v[u];
    __syncthreads();
    const uint32_t wn = w + gridDim.x;
    float2 *nxt = tile_ptr(wn < n_work ? wn : w);
    if (wn < n_work) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const vf4 x = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(nxt + uoff[u]));
        v[u] = make_float4(x.x, x.y, x.z, x.w);
      }
    }
    const float *mrow = a.mats + (size_t)b * a.G * 32;
    for (int g = 0; g < a.G; ++g) {
      const int b0 = a.gbits[g][0], b1 = a.gbits[g][1], b2 = a.gbits[g][2], b3 = a.gbits[g][3];
      uint32_t off[16];
#pragma unroll
      for (int c = 0; c < 16; ++c)
        off[c] = sw(((c & 1) ? (1u << b0) : 0u) | ((c & 2) ? (1u << b1) : 0u) |
                    ((c & 4) ? (1u << b2) : 0u) | ((c & 8) ? (1u << b3) : 0u));
      const uint32_t bs = sw(ins0(ins0(ins0(ins0(tid, b0), b1), b2), b3));
      const u64 *m = reinterpret_cast<const u64 *>(mrow + g * 32);
      u64 r[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) r[c] = reinterpret_cast<const u64 *>(s)[bs ^ off[c]];
      gate16<0>(r, m[0], m[1], m[2], m[3]);
      gate16<1>(r, m[4], m[5], m[6], m[7]);
      gate16<2>(r, m[8], m[9], m[10], m[11]);
      gate16<3>(r, m[12], m[13], m[14], m[15]);
#pragma unroll
      for (int c = 0; c < 16; ++c) reinterpret_cast<u64 *>(s)[bs ^ off[c]] = r[c];
      __syncthreads();
    }
    if (a.ro) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 o = reinterpret_cast<float4 *>(s)[sw(jl + u * 512u) >> 1];
        racc += o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 o = reinterpret_cast<float4 *>(s)[sw(jl + u * 512u) >> 1];
        const vf4 x = {o.x, o.y, o.z, o.w};
        __builtin_nontemporal_store(x, reinterpret_cast<vf4 *>(cur + uoff[u]));
      }
    }
    cur = nxt;
    __syncthreads();
  }
  if (a.ro) {
    for (int o = 32; o > 0; o >>= 1) racc += __shfl_down(racc, o, 64);
    if ((tid & 63) == 0) a.sink[(size_t)blockIdx.x * 4 + (tid >> 6)] = racc;
  }
}

int main(int argc, char **argv) {
  const int n = 24, T = 12;
  const int B = argc > 1 ? atoi(argv[1]) : 32;
  const int L = argc > 2 ? atoi(argv[2]) : 5;
  const size_t D = (size_t)1 << n;
  float2 *d;
  CHK(hipMalloc(&d, B * D * sizeof(float2)));
  std::vector<float2> h(D);
  srand(1);
  for (size_t i = 0; i < D; ++i) h[i] = make_float2((rand() % 2001 - 1000) * 2.4e-7f, (rand() % 2001 - 1000) * 2.4e-7f);
  for (int b = 0; b < B; ++b) CHK(hipMemcpy(d + b * D, h.data(), D * sizeof(float2), hipMemcpyHostToDevice));
  CHK(hipFuncSetAttribute((const void *)k_t2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  CHK(hipFuncSetAttribute((const void *)k_t2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  const int wgs = argc > 3 ? atoi(argv[3]) : 1280;
  float *dsink; CHK(hipMalloc(&dsink, (size_t)B * 4096 * 4 * sizeof(float)));
  CHK(hipFuncSetAttribute((const void *)k_t2p, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  const int modes[] = {11, 14, 15, 1, 5};
  for (int use_asm : modes)
    for (int G = 0; G <= 4; ++G) {
      Args a{};
      a.states = d; a.n = n; a.T = T; a.L = L; a.G = G; a.use_asm = use_asm >= 20 ? use_asm - 20 : use_asm; a.ro = use_asm >= 20; a.sink = dsink; a.mix_mod = getenv("MIX_MOD") ? atoi(getenv("MIX_MOD")) : 3; a.mix_mfma = getenv("MIX_MFMA") ? atoi(getenv("MIX_MFMA")) : 1;
      // tile: low L bits + the (T - L) bits from 12 upwards
      int nt = 0, no = 0;
      for (int p = 0; p < n; ++p) {
        const bool in = p < L || (p >= 12 && p < 12 + (T - L));
        if (in) a.tile_bits[nt++] = p; else a.outer_bits[no++] = p;
      }
      for (int g = 0; g < G; ++g)
        for (int j = 0; j < 4; ++j) a.gbits[g][j] = (g % 2 == 0 ? 4 : 8) + j - (g >= 2 ? 2 : 0);
      std::vector<float> hm((size_t)B * (G ? G : 1) * 32);
      for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g)
          for (int t = 0; t < 4; ++t) {
            const double th = 0.1 + 0.37 * t + 0.11 * g + 0.01 * b, ph = 0.3 * t + 0.2;
            const std::complex<double> e(std::cos(ph), std::sin(ph));
            const std::complex<double> m00 = std::cos(th), m01 = -std::sin(th) * e, m10 = std::sin(th) * std::conj(e), m11 = std::cos(th);
            float *o = &hm[((size_t)(b * G + g) * 4 + t) * 8];
            o[0] = m00.real(); o[1] = m00.imag(); o[2] = m01.real(); o[3] = m01.imag();
            o[4] = m10.real(); o[5] = m10.imag(); o[6] = m11.real(); o[7] = m11.imag();
          }
      std::vector<float> ha((size_t)B * (G ? G : 1) * 16 * 64);
      for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g) {
          std::complex<double> U[16][16];
          for (int r = 0; r < 16; ++r)
            for (int c = 0; c < 16; ++c) {
              std::complex<double> x = 1.0;
              for (int t = 0; t < 4; ++t) {
                const float *o = &hm[((size_t)(b * G + g) * 4 + t) * 8];
                const int rb = (r >> t) & 1, cb = (c >> t) & 1;
                x *= std::complex<double>(o[(rb * 2 + cb) * 2], o[(rb * 2 + cb) * 2 + 1]);
              }
              U[r][c] = x;
            }
          for (int k = 0; k < 16; ++k)
            for (int l = 0; l < 64; ++l) {
              const int i = l & 31, col = 2 * k + (l >> 5);
              const int ar = i >> 1, cr = i & 1, ac = col >> 1, cc = col & 1;
              const std::complex<double> u = U[ar][ac];
              const double val = cr == 0 ? (cc == 0 ? u.real() : -u.imag()) : (cc == 0 ? u.imag() : u.real());
              ha[(((size_t)(b * G + g) * 16) + k) * 64 + l] = (float)val;
            }
        }
      float *da;
      CHK(hipMalloc(&da, ha.size() * sizeof(float)));
      CHK(hipMemcpy(da, ha.data(), ha.size() * sizeof(float), hipMemcpyHostToDevice));
      a.amat = da;
      float *dm;
      CHK(hipMalloc(&dm, hm.size() * sizeof(float)));
      CHK(hipMemcpy(dm, hm.data(), hm.size() * sizeof(float), hipMemcpyHostToDevice));
      a.mats = dm;
      dim3 grid(1u << (n - T), B);
      const size_t lds = (size_t)8 << T;
      // correctness of one state, once per (asm, G)
      CHK(hipMemcpy(d, h.data(), D * sizeof(float2), hipMemcpyHostToDevice));
      hipEvent_t e0, e1;
      CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
      auto launch = [&]() {
        if (a.use_asm == 2) hipLaunchKernelGGL(k_t2p, dim3(wgs), dim3(256), lds, 0, a, (uint32_t)(grid.x * grid.y));
        else if (a.use_asm) hipLaunchKernelGGL(k_t2<true>, grid, dim3(256), lds, 0, a);
        else hipLaunchKernelGGL(k_t2<false>, grid, dim3(256), lds, 0, a);
      };
      launch();
      CHK(hipDeviceSynchronize());
      std::vector<float2> got(D);
      CHK(hipMemcpy(got.data(), d, D * sizeof(float2), hipMemcpyDeviceToHost));
      // host reference for state 0
      std::vector<std::complex<float>> ref(D);
      for (size_t i = 0; i < D; ++i) ref[i] = {h[i].x, h[i].y};
      for (int g = 0; g < G; ++g)
        for (int t = 0; t < 4; ++t) {
          const int p = a.tile_bits[a.gbits[g][t]];
          const float *o = &hm[((size_t)(0 * G + g) * 4 + t) * 8];
          const std::complex<float> m00(o[0], o[1]), m01(o[2], o[3]), m10(o[4], o[5]), m11(o[6], o[7]);
          const size_t S = (size_t)1 << p;
          for (size_t i = 0; i < D; ++i)
            if (!(i & S)) {
              const std::complex<float> x = ref[i], y = ref[i | S];
              ref[i] = m00 * x + m01 * y;
              ref[i | S] = m10 * x + m11 * y;
            }
        }
      double err = 0, nrm = 0;
      for (size_t i = 0; i < D; ++i) {
        err = std::max(err, (double)std::abs(ref[i] - std::complex<float>(got[i].x, got[i].y)));
        nrm = std::max(nrm, (double)std::abs(ref[i]));
      }
      for (int w = 0; w < 2; ++w) launch();
      CHK(hipEventRecord(e0));
      const int reps = 5;
      for (int r = 0; r < reps; ++r) launch();
      CHK(hipEventRecord(e1));
      CHK(hipDeviceSynchronize());
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps / B;
      printf("asm=%d L=%d G=%d (%2d dense gates): %6.1f us/state  %5.2f TB/s moved  max|err|=%.2e (max|amp| %.2e)\n",
             use_asm, L, G, 4 * G, us, 2.0 * D * 8 / us / 1e6, err, nrm);
      fflush(stdout);
      CHK(hipFree(dm));
      CHK(hipFree(da));
    }
  return 0;
}
