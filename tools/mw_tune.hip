// Meyer-Wallach read kernels, stand-alone tuning bench (round 3).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -structurizecfg-skip-uniform-regions=true \
//         tools/mw_tune.hip -o tools/mw_tune && tools/mw_tune 28
// One "read" streams the whole 2^n state once through 2^12-amplitude LDS tiles (4 contiguous low
// bits + 8 bits starting at `lo`) and reports the cross terms c_j = sum_{bit_j = 0} psi_i
// conj(psi_{i + 2^j}) of the tile's NEW bits; the first read (lo = 4: the tile is 32 KiB of
// contiguous memory) also reports the signed populations of its 12 bits and the per-row totals
// from which the populations of every other bit follow.  Variants are timed against a bare
// tile walk of the same access pattern; results are checked against a CPU sum at n = 20.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>
#include <algorithm>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef u64 __attribute__((address_space(3))) lds_u64_t;
typedef vf4 __attribute__((address_space(3))) lds_f4_t;

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}
template <bool NT> __device__ __forceinline__ vf4 ld4(const void *p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(p));
  return *reinterpret_cast<const vf4 *>(p);
}
__device__ __forceinline__ void lds_st128(uint32_t byte, vf4 v) { *(lds_f4_t *)(uintptr_t)byte = v; }
__device__ __forceinline__ vf4 lds_ld128(uint32_t byte) { return *(const lds_f4_t *)(uintptr_t)byte; }
__device__ __forceinline__ v2f lds_ld64(uint32_t byte) {
  const u64 x = *(const lds_u64_t *)(uintptr_t)byte;
  return (v2f){__uint_as_float((uint32_t)x), __uint_as_float((uint32_t)(x >> 32))};
}
__device__ __host__ __forceinline__ constexpr uint32_t sw(uint32_t e) { return e ^ (((e >> 5) & 15u) << 1); }
__device__ __forceinline__ uint32_t ins0(uint32_t i, int p) { return ((i >> p) << (p + 1)) | (i & ((1u << p) - 1u)); }

__device__ __forceinline__ void wave_sum4_dpp63(float &a, float &b, float &c, float &d) {
#define DPP4(ctrl)                                                                     \
  "v_add_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\t"          \
  "v_add_f32_dpp %2, %2, %2 " ctrl "\n\tv_add_f32_dpp %3, %3, %3 " ctrl "\n\t"
  asm volatile("s_nop 1\n\t" DPP4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                   DPP4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                       DPP4("row_half_mirror row_mask:0xf bank_mask:0xf")
                           DPP4("row_mirror row_mask:0xf bank_mask:0xf")
                               DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf")
                                   DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf")
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef DPP4
}
template <int N> __device__ __forceinline__ void wave_sums_dpp63(float (&v)[N]) {
  float pad = 0.f;
  static_for<(N + 3) / 4>([&](auto jj) {
    constexpr int j = 4 * (int)jj;
    wave_sum4_dpp63(v[j], j + 1 < N ? v[j + 1] : pad, j + 2 < N ? v[j + 2] : pad, j + 3 < N ? v[j + 3] : pad);
  });
}

// x * conj(y) accumulated into (re, im): two packed fmas
__device__ __forceinline__ void cross_acc(v2f &s, v2f x, v2f y) {
  s = __builtin_elementwise_fma(x, y.xx, s);
  s = __builtin_elementwise_fma((v2f){x.y, -x.x}, y.yy, s);
}

constexpr int kT = 12, kThreads = 256;
constexpr int kRowFirst = 48, kRowLater = 16;

struct ReadArgs {
  const float2 *state;
  float *rows;   // [n_rows][kRowFirst | kRowLater]
  int n, lo, q;  // tile = bits 0..3 + lo..lo+3 + lo2..lo2+3; 2^q tiles per workgroup
  int lo2;       // lo + 4 for one run of 8 bits
  int hiwalk;    // 1: a workgroup walks tiles 2^(n-12-q) apart (neighbouring workgroups read neighbouring tiles)
};

// byte offset of tile `t`: outer bits [4, lo) from the low bits of t, then [lo+4, lo2), then [lo2+4, n)
__device__ __forceinline__ uint64_t tile_base_bytes(uint32_t t, int lo, int lo2) {
  const uint32_t r0 = lo - 4, r1 = lo2 - lo - 4;
  const uint64_t a = (uint64_t)(t & ((1u << r0) - 1u)) << 4 |
                     (uint64_t)((t >> r0) & ((1u << r1) - 1u)) << (lo + 4) |
                     (uint64_t)(t >> (r0 + r1)) << (lo2 + 4);
  return a << 3;
}

// WORK: 0 = loads + one add per float4 (bare walk), 1 = + LDS staging and barriers, 2 = everything,
// 3 = everything but the population butterfly, 4 = everything but gather B (timing only)
template <bool FIRST, bool PREFETCH, bool NT, int WORK>
__global__ void __launch_bounds__(kThreads) k_mw_read(const ReadArgs a) {
  extern __shared__ float4 smem4[];
  const uint32_t sbo = (uint32_t)(uintptr_t)smem4;
  const uint32_t tid = threadIdx.x;
  const uint32_t jl = 2u * tid;  // local bits 1..8 from tid; 9..11 from u; bit 0 inside the float4
  const int lo = a.lo, lo2 = a.lo2;
  // in-tile byte offset of the lane's first float4 and of step u (wave-uniform)
  const uint32_t goff = ((jl & 15u) | (((jl >> 4) & 15u) << lo) | ((jl >> 8) << lo2)) << 3;
  const uint32_t ustep = 1u << (lo2 + 1 + 3);  // local bit 9 = second run's bit 1
  const char *st = reinterpret_cast<const char *>(a.state);
  const uint32_t slb = (sw(jl) << 3) + sbo;   // staging address of u = 0; u adds u << 12
  const uint32_t n_it = 1u << a.q;
  const uint32_t tile0 = a.hiwalk ? blockIdx.x : blockIdx.x << a.q;
  const uint32_t tstep = a.hiwalk ? gridDim.x : 1u;

  constexpr int NB = FIRST ? 12 : 8;           // reported bits
  v2f cr[NB];
  static_for<NB>([&](auto k) { cr[k] = (v2f){0.f, 0.f}; });
  float zin[4] = {0.f, 0.f, 0.f, 0.f}, tot = 0.f, zw[4] = {0.f, 0.f, 0.f, 0.f};
  float dummy = 0.f;

  vf4 v[8];
  auto issue = [&](uint32_t t) {
    const char *p = st + tile_base_bytes(t, lo, lo2) + goff;
    static_for<8>([&](auto u) { v[u] = ld4<NT>(p + (size_t)u * ustep); });
  };
  issue(tile0);
  for (uint32_t it = 0; it < n_it; ++it) {
    if (!PREFETCH && it) issue(tile0 + it * tstep);
    if (WORK == 0) {
      static_for<8>([&](auto u) { dummy += v[u].x + v[u].w; });
      if (PREFETCH && it + 1 < n_it) issue(tile0 + (it + 1) * tstep);
      continue;
    }
    // ---- bits held by the lane's own 8 float4: local 0 (halves of a float4) and 9, 10, 11 (u) ----
    if (WORK >= 2) {
      if (FIRST) {
        static_for<8>([&](auto u) { cross_acc(cr[0], v[u].xy, v[u].zw); });
        if (WORK != 3) {
        float pr[16];
        static_for<8>([&](auto u) {
          const v2f q0 = v[u].xy * v[u].xy, q1 = v[u].zw * v[u].zw;
          pr[2 * u] = q0.x + q0.y;
          pr[2 * u + 1] = q1.x + q1.y;
        });
        float h0 = 0.f, h1 = 0.f, h2 = 0.f, s1[8], s2[4], s3[2];
        static_for<8>([&](auto i) { s1[i] = pr[2 * i] + pr[2 * i + 1]; h0 += pr[2 * i] - pr[2 * i + 1]; });
        static_for<4>([&](auto i) { s2[i] = s1[2 * i] + s1[2 * i + 1]; h1 += s1[2 * i] - s1[2 * i + 1]; });
        static_for<2>([&](auto i) { s3[i] = s2[2 * i] + s2[2 * i + 1]; h2 += s2[2 * i] - s2[2 * i + 1]; });
        zin[0] += h0; zin[1] += h1; zin[2] += h2; zin[3] += s3[0] - s3[1];
        const float tt = s3[0] + s3[1];
        tot += tt;
        static_for<4>([&](auto j) {
          zw[j] += __uint_as_float(__float_as_uint(tt) ^ (((it >> j) & 1u) << 31));
        });
        }
      }
      static_for<3>([&](auto k) {
        constexpr int B = FIRST ? 9 + (int)k : 5 + (int)k;  // index into cr[]: local bit 9+k (later reads: new bit 5+k)
        static_for<4>([&](auto pq) {
          constexpr int lowm = (1 << k) - 1;
          constexpr int u0 = (((int)pq & ~lowm) << 1) | ((int)pq & lowm);
          cross_acc(cr[B], v[u0].xy, v[u0 | (1 << k)].xy);
          cross_acc(cr[B], v[u0].zw, v[u0 | (1 << k)].zw);
        });
      });
    }
    if (it) __syncthreads();  // previous tile's gathers done
    static_for<8>([&](auto u) { lds_st128(slb + ((uint32_t)u << 12), v[u]); });
    __syncthreads();
    if (PREFETCH && it + 1 < n_it) issue(tile0 + (it + 1) * tstep);
    if (WORK >= 2) {
      uint32_t tg = tid;
      asm volatile("" : "+v"(tg));  // keep the gather addresses out of loop-carried registers
      // gather A: 16 amplitudes over local bits gA .. gA+3 (first read: 1..4, later: 4..7)
      constexpr int gA = FIRST ? 1 : 4;
      {
        const uint32_t bs = (sw(ins0(ins0(ins0(ins0(tg, gA), gA + 1), gA + 2), gA + 3)) << 3) + sbo;
        v2f r[16];
        static_for<16>([&](auto c) { r[c] = lds_ld64(bs ^ (sw((uint32_t)c << gA) << 3)); });
        static_for<4>([&](auto t) {
          constexpr int B = FIRST ? gA + (int)t : (int)t;
          static_for<8>([&](auto pq) {
            constexpr int lowm = (1 << t) - 1;
            constexpr int c = (((int)pq & ~lowm) << 1) | ((int)pq & lowm);
            cross_acc(cr[B], r[c], r[c | (1 << t)]);
          });
        });
      }
      if (FIRST && WORK == 4) {
      } else if (FIRST) {  // gather B: local bits 5..8
        constexpr int gB = 5;
        const uint32_t bs = (sw(ins0(ins0(ins0(ins0(tg, gB), gB + 1), gB + 2), gB + 3)) << 3) + sbo;
        v2f r[16];
        static_for<16>([&](auto c) { r[c] = lds_ld64(bs ^ (sw((uint32_t)c << gB) << 3)); });
        static_for<4>([&](auto t) {
          constexpr int B = gB + (int)t;
          static_for<8>([&](auto pq) {
            constexpr int lowm = (1 << t) - 1;
            constexpr int c = (((int)pq & ~lowm) << 1) | ((int)pq & lowm);
            cross_acc(cr[B], r[c], r[c | (1 << t)]);
          });
        });
      } else {  // local bit 8 alone: 8 float4 over local bits {0, 8, 9, 10}; thread index -> 1..7, 11
        const uint32_t e0 = ((tg & 127u) << 1) | ((tg >> 7) << 11);
        const uint32_t bs = (sw(e0) << 3) + sbo;
        vf4 r[8];
        static_for<8>([&](auto c) {
          constexpr uint32_t e = (((uint32_t)c & 1u) << 8) | (((uint32_t)c >> 1) << 9);
          r[c] = lds_ld128(bs ^ (sw(e) << 3));
        });
        static_for<4>([&](auto pq) {
          cross_acc(cr[4], r[2 * pq].xy, r[2 * pq + 1].xy);
          cross_acc(cr[4], r[2 * pq].zw, r[2 * pq + 1].zw);
        });
      }
    }
  }
  if (WORK == 0) {
    if (dummy == 123.456f) a.rows[0] = dummy;
    return;
  }
  // ---- one reduction per workgroup ----
  constexpr int NV = FIRST ? 41 : 16;
  float red_v[NV];
  static_for<NB>([&](auto k) { red_v[2 * k] = cr[k].x; red_v[2 * k + 1] = cr[k].y; });
  if (FIRST) {
    // row: [0..23] cross terms of local bit b at 2b, 2b+1; [24..35] signed populations of local
    // bits 0..11; [36] total; [37..40] total signed by bit j of the tile's index in the walk
    red_v[24] = zin[0];
    static_for<8>([&](auto k) { red_v[25 + k] = ((tid >> k) & 1u) ? -tot : tot; });
    red_v[33] = zin[1]; red_v[34] = zin[2]; red_v[35] = zin[3];
    red_v[36] = tot;
    static_for<4>([&](auto j) { red_v[37 + j] = zw[j]; });
  }
  wave_sums_dpp63(red_v);
  __syncthreads();
  float *red = reinterpret_cast<float *>(smem4);
  const uint32_t lane = tid & 63u, w = tid >> 6;
  if (lane == 63) static_for<NV>([&](auto k) { red[w * NV + k] = red_v[k]; });
  __syncthreads();
  if (tid < NV) {
    const float s = red[tid] + red[NV + tid] + red[2 * NV + tid] + red[3 * NV + tid];
    a.rows[(size_t)blockIdx.x * (FIRST ? kRowFirst : kRowLater) + tid] = s;
  }
}

// ---- read + write tile pass without gates: in place (rows of 128 B in and out) vs out of place
// with the tile written as ONE contiguous 32 KiB block (the layout-permuting pass) ----
struct RwArgs {
  const float2 *in;
  float2 *out;
  int n, lo, lo2, q;
  int contig;   // 1: out[tile * 4096 + local]; 0: same addresses as the read
  int stage;    // 1: through LDS (write, barrier, read back transposed like a gather would)
};
template <bool NT, int NLD, bool PF>
__global__ void __launch_bounds__(2048 / NLD) k_rw(const RwArgs a) {
  extern __shared__ float4 smem4[];
  const uint32_t sbo = (uint32_t)(uintptr_t)smem4;
  constexpr uint32_t NTH = 2048 / NLD;       // threads; thread t, step u -> local float4 index t + u * NTH
  const uint32_t tid = threadIdx.x, jl = 2u * tid;
  const int lo = a.lo, lo2 = a.lo2;
  auto off_of = [&](uint32_t j) {  // local amplitude index (12 bits) -> byte offset inside the state
    return ((uint64_t)(j & 15u) | ((uint64_t)((j >> 4) & 15u) << lo) | ((uint64_t)(j >> 8) << lo2)) << 3;
  };
  const size_t sb = ((size_t)blockIdx.y << a.n) * 8;
  const char *st = reinterpret_cast<const char *>(a.in) + sb;
  char *so = reinterpret_cast<char *>(a.out) + sb;
  const uint32_t n_it = 1u << a.q, tile0 = blockIdx.x << a.q;
  vf4 v[NLD], w[NLD];
  auto load = [&](uint32_t t, vf4 (&r)[NLD]) {
    const uint64_t base = tile_base_bytes(t, lo, lo2);
    static_for<NLD>([&](auto u) { r[u] = ld4<NT>(st + base + off_of(jl + 2u * (uint32_t)u * NTH)); });
  };
  auto store = [&](uint32_t t, vf4 (&r)[NLD]) {
    const uint64_t base = a.contig ? ((uint64_t)t << 15) : tile_base_bytes(t, lo, lo2);
    static_for<NLD>([&](auto u) {
      const uint32_t j = jl + 2u * (uint32_t)u * NTH;
      vf4 *o = reinterpret_cast<vf4 *>(so + base + (a.contig ? (uint64_t)j << 3 : off_of(j)));
      if (NT) __builtin_nontemporal_store(r[u], o); else *o = r[u];
    });
  };
  load(tile0, v);
  for (uint32_t it = 0; it < n_it; ++it) {
    if (a.stage) {
      if (it) __syncthreads();
      static_for<NLD>([&](auto u) { lds_st128(sbo + ((sw(jl + 2u * (uint32_t)u * NTH)) << 3), v[u]); });
      __syncthreads();
      static_for<NLD>([&](auto u) { w[u] = lds_ld128(sbo + ((sw(jl + 2u * (uint32_t)u * NTH)) << 3)); });
    } else {
      static_for<NLD>([&](auto u) { w[u] = v[u]; });
    }
    static_for<NLD>([&](auto u) { w[u].x += 1.0f; });
    if (PF) {
      if (it + 1 < n_it) load(tile0 + it + 1, v);
      store(tile0 + it, w);
    } else {
      store(tile0 + it, w);
      if (it + 1 < n_it) load(tile0 + it + 1, v);
    }
  }
}

// ---- in-place read + write pass over tiles with ARBITRARY high bit positions (8 of them) ----
struct RwAnyArgs {
  float2 *st;
  int n, q;
  int pos[8];       // the tile's high bit positions, ascending
  int outer[20];    // the other positions >= 4, ascending
};
__global__ void __launch_bounds__(kThreads) k_rw_any(const RwAnyArgs a) {
  const uint32_t tid = threadIdx.x, jl = 2u * tid;
  uint64_t goff = jl & 15u;
  for (int k = 0; k < 5; ++k) goff |= (uint64_t)((jl >> (4 + k)) & 1u) << a.pos[k];
  uint64_t uo[8];
  static_for<8>([&](auto u) {
    uo[u] = ((uint64_t)((u >> 0) & 1) << a.pos[5] | (uint64_t)((u >> 1) & 1) << a.pos[6] | (uint64_t)((u >> 2) & 1) << a.pos[7]) << 3;
  });
  char *st = reinterpret_cast<char *>(a.st) + (((size_t)blockIdx.y << a.n) << 3) + (goff << 3);
  const uint32_t n_it = 1u << a.q, tile0 = blockIdx.x << a.q;
  for (uint32_t it = 0; it < n_it; ++it) {
    const uint32_t t = tile0 + it;
    uint64_t base = 0;
    for (int i = 0; i < a.n - 12; ++i) base |= (uint64_t)((t >> i) & 1u) << a.outer[i];
    base <<= 3;
    vf4 v[8];
    static_for<8>([&](auto u) { v[u] = ld4<true>(st + base + uo[u]); });
    static_for<8>([&](auto u) { v[u].x += 1.0f; });
    static_for<8>([&](auto u) { __builtin_nontemporal_store(v[u], reinterpret_cast<vf4 *>(st + base + uo[u])); });
  }
}

// ---- T = 13 (512 work items, 9 high positions IN THE GIVEN ORDER): pos[0..2] are the three bits a
// wave's instruction spans (its 8 rows of 128 B), pos[3..5] the wave index, pos[6..8] the 8 loads
struct RwAny13Args {
  float2 *st;
  int n, q;
  int pos[9];
  int outer[20];
};
__global__ void __launch_bounds__(512) k_rw_any13(const RwAny13Args a) {
  const uint32_t tid = threadIdx.x, jl = 2u * tid;
  uint64_t goff = jl & 15u;
  for (int k = 0; k < 6; ++k) goff |= (uint64_t)((jl >> (4 + k)) & 1u) << a.pos[k];
  uint64_t uo[8];
  static_for<8>([&](auto u) {
    uo[u] = ((uint64_t)((u >> 0) & 1) << a.pos[6] | (uint64_t)((u >> 1) & 1) << a.pos[7] | (uint64_t)((u >> 2) & 1) << a.pos[8]) << 3;
  });
  char *st = reinterpret_cast<char *>(a.st) + (((size_t)blockIdx.y << a.n) << 3) + (goff << 3);
  const uint32_t n_it = 1u << a.q, tile0 = blockIdx.x << a.q;
  for (uint32_t it = 0; it < n_it; ++it) {
    const uint32_t t = tile0 + it;
    uint64_t base = 0;
    for (int i = 0; i < a.n - 13; ++i) base |= (uint64_t)((t >> i) & 1u) << a.outer[i];
    base <<= 3;
    vf4 v[8];
    static_for<8>([&](auto u) { v[u] = ld4<true>(st + base + uo[u]); });
    static_for<8>([&](auto u) { v[u].x += 1.0f; });
    static_for<8>([&](auto u) { __builtin_nontemporal_store(v[u], reinterpret_cast<vf4 *>(st + base + uo[u])); });
  }
}

// ---- workgroup dispatch rate: a kernel that touches its LDS once and writes one float per workgroup
__global__ void k_dispatch_probe(float *out) {
  extern __shared__ float probe_lds[];
  probe_lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = probe_lds[(blockIdx.x + 1) % blockDim.x];
}

// read-only walk of T = 12 tiles with 8 high positions in the given order (the measuring pass's
// memory side): pos[0..2] span a wave's instruction, pos[3..4] the wave index, pos[5..7] the 8 loads
__global__ void __launch_bounds__(kThreads) k_ro_any(const RwAnyArgs a, float *sink) {
  const uint32_t tid = threadIdx.x, jl = 2u * tid;
  uint64_t goff = jl & 15u;
  for (int k = 0; k < 5; ++k) goff |= (uint64_t)((jl >> (4 + k)) & 1u) << a.pos[k];
  uint64_t uo[8];
  static_for<8>([&](auto u) {
    uo[u] = ((uint64_t)((u >> 0) & 1) << a.pos[5] | (uint64_t)((u >> 1) & 1) << a.pos[6] | (uint64_t)((u >> 2) & 1) << a.pos[7]) << 3;
  });
  const char *st = reinterpret_cast<const char *>(a.st) + (((size_t)blockIdx.y << a.n) << 3) + (goff << 3);
  const uint32_t n_it = 1u << a.q, tile0 = blockIdx.x << a.q;
  float acc = 0.f;
  for (uint32_t it = 0; it < n_it; ++it) {
    const uint32_t t = tile0 + it;
    uint64_t base = 0;
    for (int i = 0; i < a.n - 12; ++i) base |= (uint64_t)((t >> i) & 1u) << a.outer[i];
    base <<= 3;
    vf4 v[8];
    static_for<8>([&](auto u) { v[u] = ld4<true>(st + base + uo[u]); });
    static_for<8>([&](auto u) { acc += v[u].x * v[u].y + v[u].z * v[u].w; });
  }
  if (acc == 123.456f) sink[0] = acc;
}

// two tiles per iteration (they differ in outer[0]): all 16 loads first, then the stores
// MODE 0: stores A then B back to back; 1: store A, (delay), store B; 2: loads A, B interleaved per u
template <int MODE>
__global__ void __launch_bounds__(kThreads) k_rw_any2(const RwAnyArgs a) {
  const uint32_t tid = threadIdx.x, jl = 2u * tid;
  uint64_t goff = jl & 15u;
  for (int k = 0; k < 5; ++k) goff |= (uint64_t)((jl >> (4 + k)) & 1u) << a.pos[k];
  uint64_t uo[8];
  static_for<8>([&](auto u) {
    uo[u] = ((uint64_t)((u >> 0) & 1) << a.pos[5] | (uint64_t)((u >> 1) & 1) << a.pos[6] | (uint64_t)((u >> 2) & 1) << a.pos[7]) << 3;
  });
  char *st = reinterpret_cast<char *>(a.st) + (((size_t)blockIdx.y << a.n) << 3) + (goff << 3);
  const uint64_t pair_step = (uint64_t)8 << a.outer[0];
  const uint32_t n_it = 1u << a.q, tile0 = blockIdx.x << a.q;   // pairs
  for (uint32_t it = 0; it < n_it; ++it) {
    const uint32_t t = (tile0 + it) << 1;
    uint64_t base = 0;
    for (int i = 1; i < a.n - 12; ++i) base |= (uint64_t)((t >> i) & 1u) << a.outer[i];
    base <<= 3;
    vf4 v[8], w[8];
    if (MODE == 2 || MODE == 3) {
      static_for<8>([&](auto u) { v[u] = ld4<true>(st + base + uo[u]); w[u] = ld4<true>(st + base + pair_step + uo[u]); });
    } else {
      static_for<8>([&](auto u) { v[u] = ld4<true>(st + base + uo[u]); });
      static_for<8>([&](auto u) { w[u] = ld4<true>(st + base + pair_step + uo[u]); });
    }
    static_for<8>([&](auto u) { v[u].x += 1.0f; w[u].x += 1.0f; });
    if (MODE == 2 || MODE == 4) {
      static_for<8>([&](auto u) {
        __builtin_nontemporal_store(v[u], reinterpret_cast<vf4 *>(st + base + uo[u]));
        __builtin_nontemporal_store(w[u], reinterpret_cast<vf4 *>(st + base + pair_step + uo[u]));
      });
    } else {
      static_for<8>([&](auto u) { __builtin_nontemporal_store(v[u], reinterpret_cast<vf4 *>(st + base + uo[u])); });
      if (MODE == 1) __builtin_amdgcn_s_sleep(64);
      static_for<8>([&](auto u) { __builtin_nontemporal_store(w[u], reinterpret_cast<vf4 *>(st + base + pair_step + uo[u])); });
    }
  }
}

__global__ void k_init(float2 *s, uint64_t count, float scale) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint64_t x = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
  x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
  const float re = ((int)(x & 0xffff) - 32768) / 32768.f, im = ((int)((x >> 16) & 0xffff) - 32768) / 32768.f;
  // a smooth envelope so that populations and cross terms differ from bit to bit
  const float env = 1.f + 0.5f * __sinf((float)(i & 0xfffff) * 1e-5f) + 0.25f * ((i >> 7) & 1) + 0.125f * ((i >> 19) & 1);
  s[i] = make_float2(re * env * scale, (0.3f + im) * env * scale);
}

template <bool FIRST, bool PF, bool NT, int WORK>
static float run(const ReadArgs &a, int reps, const char *label) {
  const uint32_t tiles = 1u << (a.n - kT);
  const dim3 grid(tiles >> a.q);
  const size_t lds = (size_t)8 << kT;
  CK(hipFuncSetAttribute((const void *)k_mw_read<FIRST, PF, NT, WORK>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_mw_read<FIRST, PF, NT, WORK>), grid, dim3(kThreads), lds, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_mw_read<FIRST, PF, NT, WORK>), grid, dim3(kThreads), lds, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double gb = 8.0 * std::ldexp(1.0, a.n) / 1e9;
  if (label)
    printf("%-36s bits %2d-%2d,%2d-%2d q=%d hw=%d %8.4f ms  %7.1f GB/s  %.3f of 8 TB/s\n", label, a.lo, a.lo + 3, a.lo2,
           a.lo2 + 3, a.q, a.hiwalk, ms, gb / (ms * 1e-3), gb / (ms * 1e-3) / 8000.0);
  fflush(stdout);
  return ms;
}

static int check(int n) {
  const uint64_t D = 1ull << n;
  float2 *d;
  CK(hipMalloc(&d, D * sizeof(float2)));
  hipLaunchKernelGGL(k_init, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, 0, d, D, 1.0f / std::sqrt((float)D));
  std::vector<float2> h(D);
  CK(hipMemcpy(h.data(), d, D * sizeof(float2), hipMemcpyDeviceToHost));
  std::vector<double> cr(n), ci(n), z(n);
  double tot = 0;
  for (uint64_t i = 0; i < D; ++i) {
    const double p = (double)h[i].x * h[i].x + (double)h[i].y * h[i].y;
    tot += p;
    for (int j = 0; j < n; ++j) {
      if (i >> j & 1) { z[j] -= p; continue; }
      z[j] += p;
      const float2 y = h[i | (1ull << j)];
      cr[j] += (double)h[i].x * y.x + (double)h[i].y * y.y;
      ci[j] += (double)h[i].y * y.x - (double)h[i].x * y.y;
    }
  }
  int bad = 0;
  const int q = 2;
  const uint32_t tiles = 1u << (n - kT), n_rows = tiles >> q;
  float *rows;
  CK(hipMalloc(&rows, (size_t)n_rows * kRowFirst * sizeof(float)));
  std::vector<float> hr((size_t)n_rows * kRowFirst);
  // first read
  for (int pf = 0; pf < 2; ++pf) {
    ReadArgs a{d, rows, n, 4, q, 8, 0};
    if (pf) run<true, true, false, 2>(a, 1, nullptr); else run<true, false, false, 2>(a, 1, nullptr);
    CK(hipMemcpy(hr.data(), rows, hr.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<double> gcr(n), gci(n), gz(n);
    double gtot = 0;
    for (uint32_t r = 0; r < n_rows; ++r) {
      const float *row = &hr[(size_t)r * kRowFirst];
      for (int b = 0; b < 12; ++b) { gcr[b] += row[2 * b]; gci[b] += row[2 * b + 1]; gz[b] += row[24 + b]; }
      gtot += row[36];
      for (int j = 0; j < q; ++j) gz[12 + j] += row[37 + j];
      for (int j = 12 + q; j < n; ++j) gz[j] += ((r >> (j - 12 - q)) & 1) ? -row[36] : row[36];
    }
    for (int b = 0; b < n; ++b) {
      if (std::fabs(gz[b] - z[b]) > 5e-6) { printf("first pf=%d z[%d] %g vs %g\n", pf, b, gz[b], z[b]); ++bad; }
      if (b < 12 && (std::fabs(gcr[b] - cr[b]) > 5e-6 || std::fabs(gci[b] - ci[b]) > 5e-6)) {
        printf("first pf=%d c[%d] (%g, %g) vs (%g, %g)\n", pf, b, gcr[b], gci[b], cr[b], ci[b]); ++bad;
      }
    }
    if (std::fabs(gtot - tot) > 5e-6) { printf("first tot %g vs %g\n", gtot, tot); ++bad; }
  }
  // later reads: one run of 8 bits, and two runs of 4
  const int pairs[][2] = {{12, 16}, {n - 8, n - 4}, {12, n - 4}, {13, n - 5}};
  for (auto &pr : pairs) {
    const int lo = pr[0], lo2 = pr[1];
    if (lo < 12 || lo2 < lo + 4 || lo2 + 4 > n) continue;
    for (int pf = 0; pf < 4; ++pf) {
      ReadArgs a{d, rows, n, lo, q, lo2, pf >> 1};
      if (pf & 1) run<false, true, false, 2>(a, 1, nullptr); else run<false, false, false, 2>(a, 1, nullptr);
      CK(hipMemcpy(hr.data(), rows, (size_t)n_rows * kRowLater * sizeof(float), hipMemcpyDeviceToHost));
      for (int k = 0; k < 8; ++k) {
        const int bit = k < 4 ? lo + k : lo2 + k - 4;
        double r0 = 0, r1 = 0;
        for (uint32_t r = 0; r < n_rows; ++r) { r0 += hr[(size_t)r * kRowLater + 2 * k]; r1 += hr[(size_t)r * kRowLater + 2 * k + 1]; }
        if (std::fabs(r0 - cr[bit]) > 5e-6 || std::fabs(r1 - ci[bit]) > 5e-6) {
          printf("later bits %d,%d variant %d c[%d] (%g, %g) vs (%g, %g)\n", lo, lo2, pf, bit, r0, r1, cr[bit], ci[bit]); ++bad;
        }
      }
    }
  }
  printf("check n=%d: %s (tot %.9f, c[5] = %.6g %+.6gi, c[%d] = %.6g %+.6gi)\n", n, bad ? "MISMATCH" : "ok", tot, cr[5], ci[5],
         n - 1, cr[n - 1], ci[n - 1]);
  CK(hipFree(rows)); CK(hipFree(d));
  return bad;
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 28;
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  if (check(20) || check(23)) return 1;
  const uint64_t D = 1ull << n;
  float2 *d;
  CK(hipMalloc(&d, D * sizeof(float2)));
  hipLaunchKernelGGL(k_init, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, 0, d, D, 1.0f / std::sqrt((float)D));
  float *rows;
  CK(hipMalloc(&rows, ((size_t)1 << (n - kT)) * kRowFirst * sizeof(float)));
  CK(hipDeviceSynchronize());
  const int sweep = argc > 3 ? atoi(argv[3]) : 0;
  if (sweep == 2) {  // the product's sequence: later, later, first -- per-kernel HIP-event times over `reps` rounds
    const size_t lds = (size_t)8 << kT;
    ReadArgs a{d, rows, n, 4, 4, 8, 0}, b{d, rows, n, 12, 2, 24, 0}, c{d, rows, n, 16, 2, 20, 0};
    CK(hipFuncSetAttribute((const void *)k_mw_read<true, false, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void *)k_mw_read<false, false, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    std::vector<hipEvent_t> ev(4 * reps);
    for (auto &e : ev) CK(hipEventCreate(&e));
    const uint32_t tiles = 1u << (n - kT);
    for (int order = 0; order < 2; ++order) {
      for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(ev[4 * r]));
        if (order == 0) hipLaunchKernelGGL((k_mw_read<true, false, true, 2>), dim3(tiles >> a.q), dim3(kThreads), lds, 0, a);
        else hipLaunchKernelGGL((k_mw_read<false, false, true, 2>), dim3(tiles >> b.q), dim3(kThreads), lds, 0, b);
        CK(hipEventRecord(ev[4 * r + 1]));
        hipLaunchKernelGGL((k_mw_read<false, false, true, 2>), dim3(tiles >> c.q), dim3(kThreads), lds, 0, c);
        CK(hipEventRecord(ev[4 * r + 2]));
        if (order == 0) hipLaunchKernelGGL((k_mw_read<false, false, true, 2>), dim3(tiles >> b.q), dim3(kThreads), lds, 0, b);
        else hipLaunchKernelGGL((k_mw_read<true, false, true, 2>), dim3(tiles >> a.q), dim3(kThreads), lds, 0, a);
        CK(hipEventRecord(ev[4 * r + 3]));
      }
      CK(hipDeviceSynchronize());
      double t[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
      for (int r = 0; r < reps; ++r)
        for (int k = 0; k < 3; ++k) {
          float ms;
          CK(hipEventElapsedTime(&ms, ev[4 * r + k], ev[4 * r + k + 1]));
          t[k] += ms; if (ms > mx[k]) mx[k] = ms;
        }
      printf("sequence %s: %.4f / %.4f / %.4f ms avg (max %.4f / %.4f / %.4f), sum %.4f\n",
             order == 0 ? "first, later(16,20), later(12,24)" : "later(12,24), later(16,20), first", t[0] / reps, t[1] / reps,
             t[2] / reps, mx[0], mx[1], mx[2], (t[0] + t[1] + t[2]) / reps);
    }
  }
  if (sweep == 3) {  // the product library's qmle_meyer_wallach on THIS process's buffer
    void *h = dlopen(argc > 4 ? argv[4] : "qml-essentials_amd/libqmle_sv.so", RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    auto wsb = (size_t (*)(int, int))dlsym(h, "qmle_meyer_wallach_workspace_bytes");
    auto mw = (int (*)(const void *, int, int, float *, float *, void *, size_t, void *))dlsym(h, "qmle_meyer_wallach");
    const size_t wb = wsb(n, 1);
    void *ws; float *out;
    CK(hipMalloc(&ws, wb)); CK(hipMalloc(&out, 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int round = 0; round < 3; ++round) {
      int rc = mw(d, n, 1, out, nullptr, ws, wb, nullptr);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) rc |= mw(d, n, 1, out, nullptr, ws, wb, nullptr);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      float q; CK(hipMemcpy(&q, out, 4, hipMemcpyDeviceToHost));
      printf("libqmle_sv qmle_meyer_wallach: rc %d  Q %.6f  %.4f ms per call over %d calls\n", rc, q, ms / reps, reps);
    }
  }
  if (sweep == 4) {  // read + write pass, n qubits x `states` states (argv[4]) per launch
    const int states = argc > 4 ? atoi(argv[4]) : 1;
    float2 *d2;
    CK(hipFree(d));
    CK(hipMalloc(&d, ((size_t)states << n) * 8));
    CK(hipMalloc(&d2, ((size_t)states << n) * 8));
    CK(hipMemset(d, 0, ((size_t)states << n) * 8));
    CK(hipMemset(d2, 0, ((size_t)states << n) * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t tiles = 1u << (n - kT);
    const int shapes[][2] = {{4, 8}, {12, 16}};
    auto time_it = [&](auto kern, int threads, const RwArgs &a, const char *what) {
      const dim3 grid(tiles >> a.q, states);
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, grid, dim3(threads), 32768, 0, a);
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, grid, dim3(threads), 32768, 0, a);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= reps;
      const double gb = 16.0 * std::ldexp(1.0, n) * states / 1e9;
      printf("rw bits %2d-%2d,%2d-%2d q=%d %-26s %s%s %8.4f ms  %7.2f us/state  %7.1f GB/s\n", a.lo, a.lo + 3, a.lo2, a.lo2 + 3, a.q, what,
             a.contig ? "contig-out " : "in-place   ", a.stage ? "lds" : "   ", ms, ms * 1e3 / states, gb / (ms * 1e-3));
      fflush(stdout);
    };
    for (auto &sh : shapes)
      for (int q : {0, 2})
        for (int mode : {0, 2, 3}) {
          RwArgs a{d, mode == 0 ? d : d2, n, sh[0], sh[1], q, mode >= 2, mode == 3};
          time_it(k_rw<true, 8, false>, 256, a, "8 per thread, 256 thr");
          if (q) time_it(k_rw<true, 8, true>, 256, a, "8 per thread, 256 thr, pf");
          time_it(k_rw<true, 4, false>, 512, a, "4 per thread, 512 thr");
          if (q) time_it(k_rw<true, 4, true>, 512, a, "4 per thread, 512 thr, pf");
          time_it(k_rw<true, 2, false>, 1024, a, "2 per thread, 1024 thr");
          if (q) time_it(k_rw<true, 2, true>, 1024, a, "2 per thread, 1024 thr, pf");
        }
    return 0;
  }
  if (sweep == 5) {  // in-place read + write pass, arbitrary tile bits: argv[4] = states, argv[5..] = "b0,b1,..,b7" sets
    const int states = argc > 4 ? atoi(argv[4]) : 32;
    CK(hipFree(d));
    CK(hipMalloc(&d, ((size_t)states << n) * 8));
    CK(hipMemset(d, 0, ((size_t)states << n) * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int ai = 5; ai < argc; ++ai) {
      RwAnyArgs a;
      a.st = d; a.n = n; a.q = 2;
      int k = 0;
      const char *c = argv[ai];
      for (; *c && *c != ':' && k < 8;) { a.pos[k++] = atoi(c); while (*c && *c != ',' && *c != ':') ++c; if (*c == ',') ++c; }
      if (k != 8) { printf("need 8 positions: %s\n", argv[ai]); continue; }
      uint32_t used = 15u;
      for (int i = 0; i < 8; ++i) used |= 1u << a.pos[i];
      int no = 0;
      // optional ":q:o1,o2,.." = tiles per workgroup 2^q and the outer positions the LOW tile-index bits map to
      if (*c == ':') {
        ++c; a.q = atoi(c); while (*c && *c != ':') ++c;
        if (*c == ':') ++c;
        while (*c) { const int b = atoi(c); if (b >= 4 && b < n && !(used >> b & 1u)) { a.outer[no++] = b; used |= 1u << b; } while (*c && *c != ',') ++c; if (*c) ++c; }
      }
      for (int b = 4; b < n; ++b) if (!(used >> b & 1u)) a.outer[no++] = b;
      const dim3 grid((1u << (n - 12)) >> a.q, states);
      for (int variant = 0; variant < 6; ++variant) {
        const dim3 g2 = variant ? dim3(grid.x >> 1, states) : grid;
        auto launch = [&]() {
          if (variant == 0) hipLaunchKernelGGL(k_rw_any, g2, dim3(kThreads), 0, 0, a);
          else if (variant == 1) hipLaunchKernelGGL(k_rw_any2<0>, g2, dim3(kThreads), 0, 0, a);
          else if (variant == 2) hipLaunchKernelGGL(k_rw_any2<1>, g2, dim3(kThreads), 0, 0, a);
          else if (variant == 3) hipLaunchKernelGGL(k_rw_any2<2>, g2, dim3(kThreads), 0, 0, a);
          else if (variant == 4) hipLaunchKernelGGL(k_rw_any2<3>, g2, dim3(kThreads), 0, 0, a);
          else hipLaunchKernelGGL(k_rw_any2<4>, g2, dim3(kThreads), 0, 0, a);
        };
        for (int w = 0; w < 3; ++w) launch();
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        printf("rw-any {%s} %s: %7.2f us/state  %7.1f GB/s\n", argv[ai],
               variant == 0 ? "one tile per iteration      " : variant == 1 ? "pair: 16 loads, 8 + 8 stores" : variant == 2 ? "pair: stores A, sleep, B    " : variant == 3 ? "pair: interleaved per u     " : variant == 4 ? "pair: loads interleaved only" : "pair: stores interleaved only",
               ms * 1e3 / states, 16.0 * std::ldexp(1.0, n) * states / 1e9 / (ms * 1e-3));
        fflush(stdout);
      }
    }
    return 0;
  }
  if (sweep == 6) {  // T = 13 orderings: argv[4] = states, argv[5] = mode (0: listed orders, 1: search), argv[6..] = "p0,..,p8" sets
    const int states = argc > 4 ? atoi(argv[4]) : 32;
    const int mode = argc > 5 ? atoi(argv[5]) : 0;
    CK(hipFree(d));
    CK(hipMalloc(&d, ((size_t)states << n) * 8));
    CK(hipMemset(d, 0, ((size_t)states << n) * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_order = [&](const int *pos, int q) {
      RwAny13Args a;
      a.st = d; a.n = n; a.q = q;
      uint32_t used = 15u;
      for (int i = 0; i < 9; ++i) { a.pos[i] = pos[i]; used |= 1u << pos[i]; }
      int no = 0;
      for (int b = 4; b < n; ++b) if (!(used >> b & 1u)) a.outer[no++] = b;
      const dim3 grid((1u << (n - 13)) >> q, states);
      for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_rw_any13, grid, dim3(512), 0, 0, a);
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_rw_any13, grid, dim3(512), 0, 0, a);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      return ms / reps * 1e3 / states;
    };
    if (mode == 2) {  // data for the scheduler's cost model: random 9-subsets of 4..n-1, sorted + random orders
      const int n_sets = argc > 6 ? atoi(argv[6]) : 200, n_orders = argc > 7 ? atoi(argv[7]) : 4;
      uint64_t rs = 0x9E3779B97F4A7C15ull;
      auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); };
      for (int si = 0; si < n_sets; ++si) {
        int cand[32], nc = 0;
        for (int b = 4; b < n; ++b) cand[nc++] = b;
        for (int i = 0; i < 9; ++i) { const int j = i + rnd() % (nc - i); std::swap(cand[i], cand[j]); }
        int set[9];
        for (int i = 0; i < 9; ++i) set[i] = cand[i];
        std::sort(set, set + 9);
        for (int oi = 0; oi <= n_orders; ++oi) {
          int o[9];
          for (int i = 0; i < 9; ++i) o[i] = set[i];
          if (oi) for (int i = 0; i < 9; ++i) { const int j = i + rnd() % (9 - i); std::swap(o[i], o[j]); }
          const float us = time_order(o, 0);
          printf("D %d %.2f", oi, us);
          for (int i = 0; i < 9; ++i) printf(" %d", o[i]);
          printf("\n");
        }
        if (si % 20 == 0) fflush(stdout);
      }
      return 0;
    }
    for (int ai = 6; ai < argc; ++ai) {
      int set[9], k = 0;
      const char *c = argv[ai];
      for (; *c && k < 9;) { set[k++] = atoi(c); while (*c && *c != ',') ++c; if (*c == ',') ++c; }
      if (k != 9) { printf("need 9 positions: %s\n", argv[ai]); continue; }
      if (mode == 0) {
        for (int q = 0; q <= 2; ++q) printf("rw13 {%s} q=%d: %7.2f us/state\n", argv[ai], q, time_order(set, q));
        fflush(stdout);
        continue;
      }
      // search: every choice of the wave's three positions x every choice of the three load positions
      // (the remaining three index the wave); ascending inside each class; the given order is the baseline
      struct R { float us; int o[9]; };
      std::vector<R> res;
      const float base0 = time_order(set, 0);
      for (int m1 = 0; m1 < 512; ++m1) {
        if (__builtin_popcount(m1) != 3) continue;
        for (int m2 = 0; m2 < 512; ++m2) {
          if (__builtin_popcount(m2) != 3 || (m1 & m2)) continue;
          R r;
          int w = 0;
          for (int i = 0; i < 9; ++i) if (m1 >> i & 1) r.o[w++] = set[i];
          for (int i = 0; i < 9; ++i) if (!((m1 | m2) >> i & 1)) r.o[w++] = set[i];
          for (int i = 0; i < 9; ++i) if (m2 >> i & 1) r.o[w++] = set[i];
          r.us = time_order(r.o, 0);
          res.push_back(r);
        }
      }
      const float base1 = time_order(set, 0);
      std::sort(res.begin(), res.end(), [](const R &x, const R &y) { return x.us < y.us; });
      printf("set {%s}: given order %.2f / %.2f us/state; %zu orders, median %.2f, worst %.2f\n", argv[ai], base0, base1,
             res.size(), res[res.size() / 2].us, res.back().us);
      for (int i = 0; i < 12 && i < (int)res.size(); ++i) {
        printf("   %.2f :", res[i].us);
        for (int j = 0; j < 9; ++j) printf(" %d%s", res[i].o[j], j == 2 || j == 5 ? " |" : "");
        printf("\n");
      }
      // what decides: the best time per wave triple (over the load triples)
      std::vector<R> best;
      for (auto &r : res) {
        bool seen = false;
        for (auto &b : best) if (b.o[0] == r.o[0] && b.o[1] == r.o[1] && b.o[2] == r.o[2]) { seen = true; break; }
        if (!seen) best.push_back(r);
      }
      printf("   best per wave triple:");
      for (int i = 0; i < (int)best.size(); ++i) printf("%s {%d,%d,%d} %.1f", i % 8 == 0 ? "\n     " : "", best[i].o[0], best[i].o[1], best[i].o[2], best[i].us);
      printf("\n");
      fflush(stdout);
    }
    return 0;
  }
  if (sweep == 7) {  // dispatch rate: argv[4] = workgroups; block sizes 64..512 x LDS 1..64 KiB
    const int wgs = argc > 4 ? atoi(argv[4]) : 65536;
    float *o;
    CK(hipMalloc(&o, (size_t)wgs * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int threads = 64; threads <= 512; threads *= 2)
      for (int kib = 1; kib <= 64; kib *= 2) {
        const int g = wgs * 64 / threads;  // same number of waves
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_dispatch_probe, dim3(g), dim3(threads), kib * 1024, 0, o);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_dispatch_probe, dim3(g), dim3(threads), kib * 1024, 0, o);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("dispatch: %6d workgroups x %3d threads, %2d KiB LDS: %7.1f us per launch (%.1f workgroups/us, %.1f waves/us)\n", g, threads, kib,
               ms / reps * 1e3, g / (ms / reps * 1e3), (double)g * threads / 64 / (ms / reps * 1e3));
      }
    return 0;
  }
  if (sweep == 8) {  // read-only T = 12 shapes holding argv[5] ("p,q,..": forced positions), argv[4] = states, argv[6] = q
    const int states = argc > 4 ? atoi(argv[4]) : 32;
    const int q = argc > 6 ? atoi(argv[6]) : 2;
    CK(hipFree(d));
    CK(hipMalloc(&d, ((size_t)states << n) * 8));
    CK(hipMemset(d, 0, ((size_t)states << n) * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_order = [&](const int *pos) {
      RwAnyArgs a;
      a.st = d; a.n = n; a.q = q;
      uint32_t used = 15u;
      for (int i = 0; i < 8; ++i) { a.pos[i] = pos[i]; used |= 1u << pos[i]; }
      int no = 0;
      for (int b = 4; b < n; ++b) if (!(used >> b & 1u)) a.outer[no++] = b;
      const dim3 grid((1u << (n - 12)) >> q, states);
      for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_ro_any, grid, dim3(kThreads), 0, 0, a, rows);
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_ro_any, grid, dim3(kThreads), 0, 0, a, rows);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      return ms / reps * 1e3 / states;
    };
    int forced[8], nf = 0;
    if (argc > 5) { const char *c = argv[5]; for (; *c && nf < 8;) { forced[nf++] = atoi(c); while (*c && *c != ',') ++c; if (*c == ',') ++c; } }
    uint32_t fm = 0;
    for (int i = 0; i < nf; ++i) fm |= 1u << forced[i];
    struct R { float us; int o[8]; };
    std::vector<R> res;
    const int need = 8 - nf;
    for (uint32_t m = 0; m < (1u << n); m += 16) {  // candidate sets of positions 4..n-1
      if (m & fm) continue;
      if (__builtin_popcount(m) != need) continue;
      R r; int w = 0;
      for (int b = 4; b < n; ++b) if (((m | fm) >> b) & 1u) r.o[w++] = b;
      r.us = time_order(r.o);
      res.push_back(r);
    }
    std::sort(res.begin(), res.end(), [](const R &x, const R &y) { return x.us < y.us; });
    printf("read-only T=12, forced {%s}, q=%d: %zu sets (ascending order); median %.2f worst %.2f us/state\n", argc > 5 ? argv[5] : "", q, res.size(),
           res[res.size() / 2].us, res.back().us);
    for (int i = 0; i < (std::getenv("RO_ALL") ? (int)res.size() : 25) && i < (int)res.size(); ++i) {
      printf("   %.2f :", res[i].us);
      for (int j = 0; j < 8; ++j) printf(" %d", res[i].o[j]);
      printf("\n");
    }
    // orders of the best three sets: every wave triple, rest ascending
    for (int bi = 0; bi < 3 && bi < (int)res.size(); ++bi) {
      const R base = res[bi];
      std::vector<R> ord;
      for (int m1 = 0; m1 < 256; ++m1) {
        if (__builtin_popcount(m1) != 3) continue;
        R r; int w = 0;
        for (int i = 0; i < 8; ++i) if (m1 >> i & 1) r.o[w++] = base.o[i];
        for (int i = 0; i < 8; ++i) if (!(m1 >> i & 1)) r.o[w++] = base.o[i];
        r.us = time_order(r.o);
        ord.push_back(r);
      }
      std::sort(ord.begin(), ord.end(), [](const R &x, const R &y) { return x.us < y.us; });
      printf("  orders of set %d (ascending %.2f): best", bi, time_order(base.o));
      for (int i = 0; i < 4; ++i) { printf("  %.2f [", ord[i].us); for (int j = 0; j < 8; ++j) printf("%d%s", ord[i].o[j], j == 2 ? " | " : j < 7 ? "," : "]"); }
      printf("  worst %.2f\n", ord.back().us);
    }
    return 0;
  }
  if (sweep == 1) {  // sustained (thermal steady state): `reps` launches per line, the set run twice
    for (int rep = 0; rep < 2; ++rep) {
      ReadArgs a{d, rows, n, 4, 4, 8, 0};
      run<true, false, true, 0>(a, reps, "first  bare loads        nt");
      run<true, false, true, 1>(a, reps, "first  + staging         nt");
      run<true, false, true, 4>(a, reps, "first  no gather B       nt");
      run<true, false, true, 3>(a, reps, "first  no populations    nt");
      run<true, false, true, 2>(a, reps, "first  full              nt");
      ReadArgs b{d, rows, n, 12, 2, 24, 0};
      run<false, false, true, 0>(b, reps, "later  bare loads        nt");
      run<false, false, true, 1>(b, reps, "later  + staging         nt");
      run<false, false, true, 2>(b, reps, "later  full              nt");
      ReadArgs c{d, rows, n, 16, 2, 20, 0};
      run<false, false, true, 2>(c, reps, "later  full              nt");
    }
  }
  if (sweep == 0) {
    for (int q : {2, 4}) {
      ReadArgs a{d, rows, n, 4, q, 8, 0};
      run<true, false, true, 0>(a, reps, "first  bare loads        nt");
      run<true, false, true, 2>(a, reps, "first  full              nt");
      // second-run position sweep: bits 12-15 + loB..loB+3, then pairs without 12-15
      const int pairs[][2] = {{12, 16}, {12, 20}, {12, 24}, {16, 20}, {16, 24}, {20, 24}, {14, 18}, {18, 22}};
      for (auto &pr : pairs) {
        if (pr[1] + 4 > n) continue;
        for (int hw = 0; hw < 2; ++hw) {
          ReadArgs b{d, rows, n, pr[0], q, pr[1], hw};
          run<false, false, true, 0>(b, reps, "later  bare loads        nt");
          run<false, false, true, 2>(b, reps, "later  full              nt");
          run<false, true, true, 2>(b, reps, "later  full        pf    nt");
        }
      }
    }
  }
  return 0;
}
