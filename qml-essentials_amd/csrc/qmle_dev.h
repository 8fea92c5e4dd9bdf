// Device and host helpers every translation unit of libqmle_sv includes (anonymous namespace: each
// unit gets its own copy, nothing here has external linkage).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <cstdint>
#include <utility>

#include "qmle_internal.h"

using namespace qmle;

#define HIPCHK(expr)                                   \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) {                            \
      g_last_hip_error = (int)_e;                      \
      return QMLE_ERR_HIP;                             \
    }                                                  \
  } while (0)

static thread_local int g_last_hip_error = 0;

namespace {

constexpr int kWave = 64;

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ins0(uint32_t i, int p) {
  return ((i >> p) << (p + 1)) | (i & ((1u << p) - 1u));
}
__device__ __forceinline__ uint64_t ins0_64(uint64_t i, int p) {
  return ((i >> p) << (p + 1)) | (i & ((1ull << p) - 1ull));
}
// Complex multiply-add on packed fp32: a * b (+ c) is exactly two VOP3P instructions --
//   v_pk_mul/fma_f32 (a.x, a.x) * (b.x, b.y) [+ c]   and   v_pk_fma_f32 (-a.y, a.y) * (b.y, b.x) + ..
// (op_sel picks the halves; a wave-uniform `a` -- a gate matrix entry -- keeps both pairs in
// SGPRs).  Written on the two-lane vector type so that LLVM selects the packed forms; the
// scalar formulation compiled to ~7 VALU instructions per complex multiply-add.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  const v2f ar = {a.x, a.x}, ai = {-a.y, a.y}, bv = {b.x, b.y}, bs = {b.y, b.x};
  const v2f r = __builtin_elementwise_fma(ai, bs, ar * bv);
  return make_float2(r.x, r.y);
}
__device__ __forceinline__ float2 cfma(float2 a, float2 b, float2 c) {  // a*b + c
  const v2f ar = {a.x, a.x}, ai = {-a.y, a.y}, bv = {b.x, b.y}, bs = {b.y, b.x}, cv = {c.x, c.y};
  const v2f r = __builtin_elementwise_fma(ai, bs, __builtin_elementwise_fma(ar, bv, cv));
  return make_float2(r.x, r.y);
}
typedef float vf4 __attribute__((ext_vector_type(4)));
// NT: non-temporal accesses for states far larger than the 256 MiB Infinity Cache
// (measured on MI355X, tools/k1_tune.hip: +5..11 % at n = 28, harmful when cache-resident)
template <bool NT> __device__ __forceinline__ float4 ld4(const float4 *p) {
  if (NT) {
    const vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *p;
}
template <bool NT> __device__ __forceinline__ void st4(float4 *p, float4 v) {
  if (NT) {
    const vf4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<vf4 *>(p));
  } else {
    *p = v;
  }
}

__device__ __forceinline__ float norm2(float2 a) { return a.x * a.x + a.y * a.y; }

struct Mat2 {
  float2 m00, m01, m10, m11;
};
__device__ __forceinline__ Mat2 load_mat2(const float *__restrict__ m) {
  Mat2 r;
  r.m00 = make_float2(m[0], m[1]);
  r.m01 = make_float2(m[2], m[3]);
  r.m10 = make_float2(m[4], m[5]);
  r.m11 = make_float2(m[6], m[7]);
  return r;
}
__device__ __forceinline__ void apply2(const Mat2 &m, float2 &a0, float2 &a1) {
  const float2 b0 = cfma(m.m01, a1, cmul(m.m00, a0));
  const float2 b1 = cfma(m.m11, a1, cmul(m.m10, a0));
  a0 = b0;
  a1 = b1;
}

typedef unsigned long long u64;
// LDS access by byte offset (address space 3: the offset IS the address -- no 64-bit generic
// pointer arithmetic, no `base + offset` add per access)
typedef u64 __attribute__((address_space(3))) lds_u64_t;
typedef float f4n_t __attribute__((ext_vector_type(4)));
typedef f4n_t __attribute__((address_space(3))) lds_f4_t;
__device__ __forceinline__ float4 lds_ld128(uint32_t byte) {
  const f4n_t v = *(const lds_f4_t *)(uintptr_t)byte;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lds_st128(uint32_t byte, const float4 &v) {
  const f4n_t w = {v.x, v.y, v.z, v.w};
  *(lds_f4_t *)(uintptr_t)byte = w;
}
__device__ __forceinline__ u64 lds_ld64(uint32_t byte) { return *(const lds_u64_t *)(uintptr_t)byte; }
__device__ __forceinline__ void lds_st64(uint32_t byte, u64 v) { *(lds_u64_t *)(uintptr_t)byte = v; }
__device__ __forceinline__ uint32_t lds_offset_of(const void *p) {  // low half of a generic LDS address
  return (uint32_t)(uintptr_t)p;
}

// x conj(y) accumulated into (re, im): two packed fmas
__device__ __forceinline__ void mw_cross(v2f &s, v2f x, v2f y) {
  s = __builtin_elementwise_fma(x, y.xx, s);
  s = __builtin_elementwise_fma((v2f){x.y, -x.x}, y.yy, s);
}

// LDS layout of a tile: amplitude e lives in slot sw(e).  XOR-ing bits 1..4 with bits
// 5..8 keeps (even, odd) pairs adjacent (float4 staging) and spreads the 16-amplitude
// register gathers of low-bit groups over the banks (<= 2-way instead of 16-way).
__device__ __forceinline__ uint32_t sw(uint32_t e) { return e ^ (((e >> 5) & 15u) << 1); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}
// Wave sum on the DPP data path (no LDS crossbar): quad swaps, half-row / row mirrors, then the
// row broadcasts; the total lands in lane 63.  One v_add_f32_dpp per step -- six per value, and
// independent values interleave freely.
__device__ __forceinline__ float wave_sum_dpp63(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false)); // row_bcast:15 -> rows 1, 3
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false)); // row_bcast:31 -> rows 2, 3
  return v;
}
// Four wave sums at once, written out: one v_add_f32_dpp per step and value (24 instructions).
// hipcc turns the builtin form above into v_mov_b32_dpp + v_pk_add_f32 pairs and materialises a
// zero per masked row broadcast -- about twice the instructions (seen in the measuring
// epilogue of k_tile2: 140 for 11 values).  Stage-major order keeps three independent
// instructions between a write and the DPP read of it (the hazard needs two).
__device__ __forceinline__ void wave_sum4_dpp63(float &a, float &b, float &c, float &d) {
#define QMLE_DPP4(ctrl)                                                                          \
  "v_add_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\t"                   \
  "v_add_f32_dpp %2, %2, %2 " ctrl "\n\tv_add_f32_dpp %3, %3, %3 " ctrl "\n\t"
  asm volatile("s_nop 1\n\t"
               QMLE_DPP4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
               QMLE_DPP4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
               QMLE_DPP4("row_half_mirror row_mask:0xf bank_mask:0xf")
               QMLE_DPP4("row_mirror row_mask:0xf bank_mask:0xf")
               QMLE_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf")
               QMLE_DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf")
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef QMLE_DPP4
}
// N values (padded to a multiple of four with a dummy)
template <int N> __device__ __forceinline__ void wave_sums_dpp63(float (&v)[N]) {
  float pad = 0.f;
#pragma unroll
  for (int j = 0; j < N; j += 4)
    wave_sum4_dpp63(v[j], j + 1 < N ? v[j + 1] : pad, j + 2 < N ? v[j + 2] : pad, j + 3 < N ? v[j + 3] : pad);
}
// Sum over the block; result valid in thread 0.  `red` holds >= 16 floats.
__device__ __forceinline__ float block_sum(float v, float *red) {
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  const int nw = (blockDim.x + kWave - 1) / kWave;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// Wave-level reduce-scatter of N <= 64 per-lane values: afterwards lane l holds the wave total of
// value l (lanes >= N hold garbage-free zeros).  Step with mask m: every lane keeps the half of
// the remaining index range selected by its own lane bit m and adds the partner's copy of it --
// 32+16+8+4+2+1 = 63 exchanges for any N <= 64, instead of 6 per value for N separate wave sums.
template <int N>
__device__ __forceinline__ float wave_reduce_scatter(const float (&v)[N]) {
  static_assert(N >= 1 && N <= 64, "at most one value per lane");
  const int lane = threadIdx.x & (kWave - 1);
  float a[32];
  {
    const bool up = lane & 32;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const float lo = j < N ? v[j] : 0.f, hi = j + 32 < N ? v[j + 32] : 0.f;
      if (j + 32 < N) a[j] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, 32, kWave);
      else if (j < N) a[j] = (up ? 0.f : lo) + __shfl_xor(up ? lo : 0.f, 32, kWave);
      else a[j] = 0.f;
    }
  }
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) {
    const bool up = lane & m;
#pragma unroll
    for (int j = 0; j < m; ++j) {
      const float lo = a[j], hi = a[j + m];
      a[j] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, m, kWave);
    }
  }
  return a[0];
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}
// fp64 block sum for the tiny "final" kernels; result valid in thread 0; red >= 16 doubles
__device__ __forceinline__ double block_sum_d(double v, double *red) {
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  const int nw = (blockDim.x + kWave - 1) / kWave;
  v = wave_sum_d(v);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// Plan data and per-sample matrices are written before the launch and never during it: reading
// them through the constant address space lets wave-uniform accesses compile to scalar loads
// (s_load_dwordx4/x8/x16 into SGPRs) instead of vector loads + v_readfirstlane.
#define QMLE_CONSTANT __attribute__((address_space(4)))
template <class X>
__device__ __forceinline__ const X QMLE_CONSTANT *as_constant(const X *p) {
  return (const X QMLE_CONSTANT *)(uintptr_t)p;
}

// compile-time loop: f(std::integral_constant<int, 0>) ... f(<N-1>).  Register arrays indexed this
// way are split into scalars by the first SROA run; arrays walked by `#pragma unroll` loops are
// turned into one wide vector value first (AMDGPU alloca-to-vector promotion) and copied around.
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}
#define QMLE_X16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// ---------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// exact grids by default: grid-stride persistence measured slower for pure streaming
inline unsigned grid_for(uint64_t items, unsigned block, unsigned cap = 1u << 30) {
  uint64_t g = (items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

// hipFuncSetAttribute (160 KiB dynamic LDS) and the CU count are per DEVICE: a process that
// drives several GPUs (one process per GPU is the supported layout, but nothing stops a caller)
// must set them on each.  `slot` = a distinct small integer per call site.
constexpr int kMaxDevices = 64;
static inline int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  return dev;
}
// `if (FirstUse once{slot}; once.first) { ...set attributes, may return an error...; once.done(); }`:
// the slot is marked done only AFTER the guarded work succeeded (an early error return leaves it to the
// next call instead of latching a half-configured device: ADVICE r3), and a second thread that arrives
// meanwhile waits on the slot's mutex instead of launching before the attributes are set.
struct FirstUse {
  std::atomic<bool> *flag = nullptr;
  std::mutex *mu = nullptr;
  bool first = true;
  explicit FirstUse(int slot) {
    static std::atomic<bool> done_[8][kMaxDevices] = {};
    static std::mutex mu_[8][kMaxDevices];
    int dev = -1;
    // a device index beyond the table has no slot of its own: its attributes are simply set on
    // every call (idempotent) instead of sharing -- and trusting -- slot 0's flag
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return;
    flag = &done_[slot][dev];
    if (flag->load(std::memory_order_acquire)) { first = false; return; }
    mu = &mu_[slot][dev];
    mu->lock();
    if (flag->load(std::memory_order_acquire)) {  // somebody else finished while we waited
      first = false;
      mu->unlock();
      mu = nullptr;
    }
  }
  FirstUse(const FirstUse &) = delete;
  FirstUse &operator=(const FirstUse &) = delete;
  void done() {
    if (flag) flag->store(true, std::memory_order_release);
  }
  ~FirstUse() {
    if (mu) mu->unlock();
  }
};

// The tile kernels address their LDS tile by XOR (swizzle + gather offsets folded into one
// `base ^ offset`), which is only an addition while the tile starts at a multiple of its size.
// The dynamic-LDS window starts right behind a kernel's static __shared__ variables, so the
// invariant is "these kernels have none": checked here on the host at first use (a static
// __shared__ added later turns into QMLE_ERR_INTERNAL at the first launch instead of wrong
// amplitudes or a device-side abort); tests/test_abi_cpu.py checks the build's resource table.
static int lds_base_is_zero(const void *kernel) {
  hipFuncAttributes attr;
  if (hipFuncGetAttributes(&attr, kernel) != hipSuccess) return QMLE_ERR_HIP;
  return attr.sharedSizeBytes == 0 ? QMLE_OK : QMLE_ERR_INTERNAL;
}
#define QMLE_LDS_BASE_CHECK(kernel)                                     \
  do {                                                                  \
    const int rc_lds_ = lds_base_is_zero((const void *)(kernel));       \
    if (rc_lds_ != QMLE_OK) return rc_lds_;                             \
  } while (0)

}  // namespace
