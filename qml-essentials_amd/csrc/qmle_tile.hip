// libqmle_sv, tile passes: gfx950 (MI355X / CDNA4) LDS-tile kernels and their launchers.
//
// Replaces the compute of qml_essentials/simulation.py (simulate_pure :65-104,
// measure_state :204-271) and the vmap batch dispatch of script.py:399-553.
// Written for CDNA4 only: 64-wide wavefronts, 160 KiB LDS per CU, 16-byte
// (float4 = 2 amplitudes) global accesses everywhere, one workgroup per LDS tile.
//
// Kernels
//   k_build_matrices   per-sample 2x2 / 4x4 gate matrices from the angle table
//   k_tile             load 2^T-amplitude tile -> LDS, apply a list of gates,
//                      store / measure   (whole state in LDS when n <= 14)
//   k_direct_1q        one (controlled) 2x2 gate streamed through HBM in place
//   k_diag_all         full-register diagonal (Golomb encoding)
//   k_reg_measure<FOLD>, k_reg_measure_mono (+ k_mono_coef)
//                      measuring last pass in registers: <Z> / Z parities accumulated
//                      across tiles per work item, gates on known zeros folded away
//   k_product_stream, k_tile_product (+ k_fold_columns)
//                      pass whose gate groups all act on known-zero bits:
//                      out = in (x) prod_g U_g e_0, written without staging amplitudes
//   k_expval_partial / k_expval_final   all-qubit <Z> in ONE read of the state
//   k_probs, k_density, k_marginal, k_overlap_*, k_cross_*, k_histogram
//   k_mw_tile*         Meyer-Wallach purities; k_adjoint_lds, k_tile_adj, k_adj_*: adjoint
//   k_cdf, k_sample, k_probs_diag_expval: shot sampling; k_build_angles: device angle table
//
// Runs from |0..0> track the bit positions whose amplitudes are still exactly zero
// (Stage::zero_in, qmle_plan.cpp): they are neither read nor computed nor stored.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <new>
#include <utility>

#include "qmle_internal.h"
#include "qmle_host.h"
#include "qmle_dev.h"
#include "qmle_tile_dev.h"

namespace {

// One tile per workgroup: load -> gate groups -> store / measure.
// MW: a.meas is TM_STORE_MW / TM_MW_ONLY (tile_mw_row behind the store; T >= 10, 2^(T-4) work items)
template <bool DENSE4, bool MW = false>
__global__ void k_tile(const TileArgs a) {
  extern __shared__ float4 smem4[];
  float2 *s = reinterpret_cast<float2 *>(smem4);
  const int T = a.T, L = a.L;
  uint32_t *lut = reinterpret_cast<uint32_t *>(s + (1u << T));
  const uint32_t lut_n = (1u << (T - L)) < 4u ? 4u : (1u << (T - L));
  float *red = reinterpret_cast<float *>(lut + lut_n);
  OpSlot *slots = reinterpret_cast<OpSlot *>(red + 288);
  const int tid = threadIdx.x, nt = blockDim.x;
  const int b = blockIdx.y;
  uint32_t tile = blockIdx.x;
  if (a.compact) {  // blockIdx.x enumerates the tiles that can be non-zero
    uint32_t rest = tile, free_bits = a.tile_free;
    tile = 0;
    while (rest) {
      const uint32_t low = free_bits & (0u - free_bits);
      if (rest & 1u) tile |= low;
      free_bits ^= low;
      rest >>= 1;
    }
  }
  const size_t D = (size_t)1 << a.n;

  const uint64_t base = tile_base(a, tile);
  tile_build_lut(a, lut);
  const uint32_t half = 1u << (T - 1);
  const uint32_t lowmask = (1u << L) - 1u;
  float2 *st = a.states + (size_t)b * D;
  if (a.init_zero ? base != 0 : (tile & a.zin_outer) != 0) {
    // |0..0> lives in tile 0 alone and gates are linear: a tile that holds only known zeros
    // stays exactly zero -- write the zeros (state / probabilities / partial sums), skip the gates
    __syncthreads();
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MW) {
      float *po = reinterpret_cast<float *>(a.out) + ((size_t)b * gridDim.x + tile) * kMwFusedRow;
      if (tid < kMwFusedRow) po[tid] = 0.f;
    }
    if (a.meas == TM_STORE || a.meas == TM_STORE_MW) {
      for (uint32_t jc = tid; jc < half; jc += nt) {
        const uint32_t j = jc * 2u;
        *reinterpret_cast<float4 *>(st + (base | lut[j >> L] | (j & lowmask))) = z;
      }
    } else if (a.meas == TM_PROBS) {
      float *po = reinterpret_cast<float *>(a.out) + (size_t)b * D;
      for (uint32_t jc = tid; jc < half; jc += nt) {
        const uint32_t j = jc * 2u;
        *reinterpret_cast<float2 *>(po + (base | lut[j >> L] | (j & lowmask))) = make_float2(0.f, 0.f);
      }
    } else if (!MW) {  // TM_EXPVAL_PARTIAL / TM_EXPVAL_MASKS rows (TM_EXPVAL has a single tile)
      float *po = reinterpret_cast<float *>(a.out) +
                  ((size_t)b * gridDim.x + tile) * (QMLE_MAX_QUBITS + 1);
      if (tid <= QMLE_MAX_QUBITS) po[tid] = 0.f;
    }
    return;
  }
  if (a.slots_in_lds) tile_stage_slots(a, slots, b);
  __syncthreads();

  if (a.init_zero) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (uint32_t jc = tid; jc < half; jc += nt) reinterpret_cast<float4 *>(s)[jc] = z;
    __syncthreads();
    if (tid == 0 && base == 0) s[sw(0)] = make_float2(1.f, 0.f);  // |0...0>, simulation.py:100
  } else if (a.zin_local) {
    // only the amplitudes that can be non-zero are read; the rest of the tile is zero-filled
    const uint32_t zl = a.zin_local & ~1u;
    const bool z0 = (a.zin_local & 1u) != 0;
    for (uint32_t jc = tid; jc < half; jc += nt) {
      const uint32_t j = jc * 2u;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((j & zl) == 0) {
        v = *reinterpret_cast<const float4 *>(st + (base | lut[j >> L] | (j & lowmask)));
        if (z0) v.z = v.w = 0.f;
      }
      reinterpret_cast<float4 *>(s)[sw(j) >> 1] = v;
    }
  } else {
    // stage the tile through registers, 8 independent 16-byte loads in flight per lane
    if ((half % (8u * nt)) == 0) {
      for (uint32_t j0 = tid; j0 < half; j0 += 8u * nt) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const uint32_t j = (j0 + u * nt) * 2u;
          const float4 *src = reinterpret_cast<const float4 *>(st + (base | lut[j >> L] | (j & lowmask)));
          v[u] = a.nt ? ld4<true>(src) : ld4<false>(src);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          reinterpret_cast<float4 *>(s)[sw((j0 + u * nt) * 2u) >> 1] = v[u];
      }
    } else {
      for (uint32_t jc = tid; jc < half; jc += nt) {
        const uint32_t j = jc * 2u;
        reinterpret_cast<float4 *>(s)[sw(j) >> 1] =
            *reinterpret_cast<const float4 *>(st + (base | lut[j >> L] | (j & lowmask)));
      }
    }
  }
  __syncthreads();

  tile_compute<DENSE4, false>(a, s, slots, b);
  if (MW) {
    if (a.meas == TM_STORE_MW) tile_epilogue<false>(a, s, lut, red, tile, gridDim.x, b, base);  // (its TM_STORE branch)
    tile_mw_row(lds_offset_of(s), T, (uint32_t)tid, reinterpret_cast<float *>(s),
                reinterpret_cast<float *>(a.out) + ((size_t)b * gridDim.x + tile) * kMwFusedRow, a.mw_lean != 0);
  } else {
    tile_epilogue<false, false, true>(a, s, lut, red, tile, gridDim.x, b, base);
  }
}

// ---- fast tile kernel (k_tile2) --------------------------------------------------------------
// Dense tile pass for stages whose gates are all (<= 1 control) 2x2 (Stage::fast_ok):
//   * the groups' LDS addresses come from host-built tables (Group2): X / CX between groups are
//     GF(2)-affine index maps folded into those tables and cost nothing (qmle_plan.cpp);
//   * gate matrices are read into SGPRs with scalar loads straight from the per-sample matrix
//     row (no LDS staging, no v_readfirstlane), and a gate on 16 amplitudes is 64 packed-fp32
//     instructions written in asm: 4 independent dependency chains interleaved, so the packed
//     pipe never waits on its own result (hipcc serialises each chain behind s_nop);
//   * no lookup table in LDS: the 8 float4 of a lane differ in wave-uniform high bits only, so
//     a tile of 2^12 amplitudes needs exactly 32 KiB -> 5 workgroups per CU.

// (b0, b1) = M (a0, a1) for two amplitude pairs under the same 2x2 matrix; complex products as
// 2 packed instructions each: (m.x, m.x) * (a.x, a.y), then (-m.y, m.y) * (a.y, a.x) + ...
#define QMLE_PAIR2(a0, a1, a2, a3)                                                               \
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
// four amplitudes times the same complex number
#define QMLE_CMUL4(a0, a1, a2, a3, m)                                                            \
  asm volatile(                                                                                  \
      "v_pk_mul_f32 %4, %8, %0 op_sel_hi:[0,1]\n\t"                                              \
      "v_pk_mul_f32 %5, %8, %1 op_sel_hi:[0,1]\n\t"                                              \
      "v_pk_mul_f32 %6, %8, %2 op_sel_hi:[0,1]\n\t"                                              \
      "v_pk_mul_f32 %7, %8, %3 op_sel_hi:[0,1]\n\t"                                              \
      "v_pk_fma_f32 %0, %8, %0, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"          \
      "v_pk_fma_f32 %1, %8, %1, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"          \
      "v_pk_fma_f32 %2, %8, %2, %6 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"          \
      "v_pk_fma_f32 %3, %8, %3, %7 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"          \
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)       \
      : "s"(m))

struct Mat2S {  // a 2x2 complex matrix as four (re, im) SGPR pairs
  u64 m00, m01, m10, m11;
};

// The 16 amplitudes of a work item: 16 named scalars, indexed at compile time only (at<I>).
// (An array walked by unrolled loops is turned into one <16 x i64> value by the AMDGPU
// alloca-to-vector promotion before the loops are unrolled; every gate then copies the whole
// 32-register tuple in and out.)
struct A16 {
  u64 v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15;
};
template <int I> __device__ __forceinline__ u64 &at(A16 &a) {
  static_assert(I >= 0 && I < 16, "amplitude index");
  if constexpr (I == 0) return a.v0; else if constexpr (I == 1) return a.v1;
  else if constexpr (I == 2) return a.v2; else if constexpr (I == 3) return a.v3;
  else if constexpr (I == 4) return a.v4; else if constexpr (I == 5) return a.v5;
  else if constexpr (I == 6) return a.v6; else if constexpr (I == 7) return a.v7;
  else if constexpr (I == 8) return a.v8; else if constexpr (I == 9) return a.v9;
  else if constexpr (I == 10) return a.v10; else if constexpr (I == 11) return a.v11;
  else if constexpr (I == 12) return a.v12; else if constexpr (I == 13) return a.v13;
  else if constexpr (I == 14) return a.v14; else return a.v15;
}

// pair q (0..7) of target bit TB: q with a zero inserted at bit TB
template <int TB> __device__ __forceinline__ constexpr int pair_idx(int q) {
  return ((q & ~((1 << TB) - 1)) << 1) | (q & ((1 << TB) - 1));
}
template <int TB>
__device__ __forceinline__ void f_dense(A16 &a, const Mat2S &M) {
  const u64 m00 = M.m00, m01 = M.m01, m10 = M.m10, m11 = M.m11;
  u64 t0, t1, t2, t3;
  constexpr int S = 1 << TB;
#define QMLE_P(q) at<pair_idx<TB>(q)>(a), at<pair_idx<TB>(q) | S>(a)
#define QMLE_PAIR2X(...) QMLE_PAIR2(__VA_ARGS__)
  QMLE_PAIR2X(QMLE_P(0), QMLE_P(1));
  QMLE_PAIR2X(QMLE_P(2), QMLE_P(3));
  QMLE_PAIR2X(QMLE_P(4), QMLE_P(5));
  QMLE_PAIR2X(QMLE_P(6), QMLE_P(7));
#undef QMLE_PAIR2X
#undef QMLE_P
}
// index r (0..3) deposited into the two bits that are neither CB nor TB, control bit set
template <int CB, int TB>
__device__ __forceinline__ constexpr int ctl_idx(int r) {
  int c = 0, k = 0;
  for (int j = 0; j < 4; ++j) {
    if (j == CB) c |= 1 << j;
    else if (j != TB) c |= ((r >> k++) & 1) << j;
  }
  return c;
}
template <int CB, int TB>
__device__ __forceinline__ void f_cdense(A16 &a, const Mat2S &M) {
  const u64 m00 = M.m00, m01 = M.m01, m10 = M.m10, m11 = M.m11;
  u64 t0, t1, t2, t3;
  constexpr int S = 1 << TB;
  constexpr int i0 = ctl_idx<CB, TB>(0), i1 = ctl_idx<CB, TB>(1), i2 = ctl_idx<CB, TB>(2),
                i3 = ctl_idx<CB, TB>(3);
  QMLE_PAIR2(at<i0>(a), at<i0 | S>(a), at<i1>(a), at<i1 | S>(a));
  QMLE_PAIR2(at<i2>(a), at<i2 | S>(a), at<i3>(a), at<i3 | S>(a));
}
template <int TB>
__device__ __forceinline__ void f_diag(A16 &a, const Mat2S &M) {
  const u64 m00 = M.m00, m11 = M.m11;
  u64 t0, t1, t2, t3;
  constexpr int S = 1 << TB;
#define QMLE_I(q) pair_idx<TB>(q)
  QMLE_CMUL4(at<QMLE_I(0)>(a), at<QMLE_I(1)>(a), at<QMLE_I(2)>(a), at<QMLE_I(3)>(a), m00);
  QMLE_CMUL4(at<QMLE_I(4)>(a), at<QMLE_I(5)>(a), at<QMLE_I(6)>(a), at<QMLE_I(7)>(a), m00);
  QMLE_CMUL4(at<QMLE_I(0) | S>(a), at<QMLE_I(1) | S>(a), at<QMLE_I(2) | S>(a), at<QMLE_I(3) | S>(a), m11);
  QMLE_CMUL4(at<QMLE_I(4) | S>(a), at<QMLE_I(5) | S>(a), at<QMLE_I(6) | S>(a), at<QMLE_I(7) | S>(a), m11);
#undef QMLE_I
}
template <int CB, int TB>
__device__ __forceinline__ void f_cdiag(A16 &a, const Mat2S &M) {
  const u64 m00 = M.m00, m11 = M.m11;
  u64 t0, t1, t2, t3;
  constexpr int S = 1 << TB;
  constexpr int i0 = ctl_idx<CB, TB>(0), i1 = ctl_idx<CB, TB>(1), i2 = ctl_idx<CB, TB>(2),
                i3 = ctl_idx<CB, TB>(3);
  QMLE_CMUL4(at<i0>(a), at<i1>(a), at<i2>(a), at<i3>(a), m00);
  QMLE_CMUL4(at<i0 | S>(a), at<i1 | S>(a), at<i2 | S>(a), at<i3 | S>(a), m11);
}
// A swap as three real moves, in place: as a renaming (t = a; a = b; b = t in C++) it is free in
// the X / CX cases but makes every amplitude's register depend on the case taken, and the joins
// of the gate switch then cost ~21 v_mov_b64 per gate on EVERY path (measured: 419 instead of 292
// vector instructions for a group of four dense gates).  X / CX inside a group are rare (most are
// folded into the LDS layout), the dense cases are what the loop runs.
template <int I, int J> __device__ __forceinline__ void swap_amp(A16 &a) {
  u64 t;
  asm volatile("v_mov_b64 %2, %0\n\tv_mov_b64 %0, %1\n\tv_mov_b64 %1, %2"
               : "+v"(at<I>(a)), "+v"(at<J>(a)), "=&v"(t));
}
template <int TB>
__device__ __forceinline__ void f_x(A16 &a) {
  constexpr int S = 1 << TB;
  swap_amp<pair_idx<TB>(0), pair_idx<TB>(0) | S>(a); swap_amp<pair_idx<TB>(1), pair_idx<TB>(1) | S>(a);
  swap_amp<pair_idx<TB>(2), pair_idx<TB>(2) | S>(a); swap_amp<pair_idx<TB>(3), pair_idx<TB>(3) | S>(a);
  swap_amp<pair_idx<TB>(4), pair_idx<TB>(4) | S>(a); swap_amp<pair_idx<TB>(5), pair_idx<TB>(5) | S>(a);
  swap_amp<pair_idx<TB>(6), pair_idx<TB>(6) | S>(a); swap_amp<pair_idx<TB>(7), pair_idx<TB>(7) | S>(a);
}
template <int CB, int TB>
__device__ __forceinline__ void f_cx(A16 &a) {
  constexpr int S = 1 << TB;
  swap_amp<ctl_idx<CB, TB>(0), ctl_idx<CB, TB>(0) | S>(a); swap_amp<ctl_idx<CB, TB>(1), ctl_idx<CB, TB>(1) | S>(a);
  swap_amp<ctl_idx<CB, TB>(2), ctl_idx<CB, TB>(2) | S>(a); swap_amp<ctl_idx<CB, TB>(3), ctl_idx<CB, TB>(3) | S>(a);
}

// one op of a Group2 on the 16 amplitudes a thread holds; `code` is wave-uniform (FastCode)
__device__ __forceinline__ void fast_dispatch(A16 &a, int code, const Mat2S &M) {
#define QMLE_C12(F, base, ...)                                                                   \
  case base + 0: F<0, 1>(__VA_ARGS__); break; case base + 1: F<0, 2>(__VA_ARGS__); break;        \
  case base + 2: F<0, 3>(__VA_ARGS__); break; case base + 3: F<1, 0>(__VA_ARGS__); break;        \
  case base + 4: F<1, 2>(__VA_ARGS__); break; case base + 5: F<1, 3>(__VA_ARGS__); break;        \
  case base + 6: F<2, 0>(__VA_ARGS__); break; case base + 7: F<2, 1>(__VA_ARGS__); break;        \
  case base + 8: F<2, 3>(__VA_ARGS__); break; case base + 9: F<3, 0>(__VA_ARGS__); break;        \
  case base + 10: F<3, 1>(__VA_ARGS__); break; case base + 11: F<3, 2>(__VA_ARGS__); break;
  // (the uncontrolled dense gate is what deep circuits are made of: two scalar branches to reach
  // it instead of the six of a balanced tree over all 48 codes)
  if (code < FC_CDENSE) {
    if (code < 2) { if (code == 0) f_dense<0>(a, M); else f_dense<1>(a, M); }
    else { if (code == 2) f_dense<2>(a, M); else f_dense<3>(a, M); }
    return;
  }
  switch (code) {
    QMLE_C12(f_cdense, FC_CDENSE, a, M)
    case FC_DIAG + 0: f_diag<0>(a, M); break;
    case FC_DIAG + 1: f_diag<1>(a, M); break;
    case FC_DIAG + 2: f_diag<2>(a, M); break;
    case FC_DIAG + 3: f_diag<3>(a, M); break;
    QMLE_C12(f_cdiag, FC_CDIAG, a, M)
    case FC_X + 0: f_x<0>(a); break;
    case FC_X + 1: f_x<1>(a); break;
    case FC_X + 2: f_x<2>(a); break;
    case FC_X + 3: f_x<3>(a); break;
    QMLE_C12(f_cx, FC_CX, a)
    default: break;
  }
#undef QMLE_C12
}


struct Tile2Args {
  const Group2 *groups;     // this stage's Group2 range
  const LoweredOp *ops;     // qmle_plan::ops2 on the device
  const uint32_t *tbl;      // qmle_plan::tbl2 on the device
  int n_groups;
  int n_ops_stage;          // ops of all the stage's groups (one contiguous stream in `ops`)
  int dbg;                  // QMLE_DBG_T2 (timing anatomy only): 1 no groups, 2 no epilogue, 4 / 8: see tile2_groups,
                            // 16 no global stores of a storing pass, 32 no global loads (constants instead)
  uint32_t gtab;            // index into tbl: per-lane global byte offset inside the tile
  uint32_t uoff8[8];        // byte offsets of the lane's 8 float4 (the tile's top three bits)
  // tile index -> amplitude offset of the tile: the outer bit positions as <= 6 contiguous runs
  // (base = sum_r ((tile >> run_off[r]) & run_mask[r]) << run_pos[r]); n_runs < 0: generic loop
  int n_runs;
  uint32_t run_off[6], run_mask[6], run_pos[6];
  // the same for the lane's own offset: local bits 1 .. T-4 of index 2 tid -> global positions,
  // <= 4 runs (n_in_runs < 0: read it from the table at gtab)
  int n_in_runs;
  uint32_t in_off[4], in_mask[4], in_pos[4];
  int tpw;                  // consecutive tiles per workgroup (plain all-live stages; else 1)
  uint32_t tile_stride;     // amplitudes between consecutive tiles of a workgroup (2^lowest outer bit)
};

__device__ __forceinline__ uint64_t tile2_base(const TileArgs &a, const Tile2Args &f, uint32_t tile) {
  if (f.n_runs < 0) return tile_base(a, tile);
  uint64_t base = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r)
    if (r < f.n_runs) base |= (uint64_t)((tile >> f.run_off[r]) & f.run_mask[r]) << f.run_pos[r];
  return base;
}


// All register-tile groups of a stage on the tile in LDS (k_tile2); `addr` = this work item's
// table entry of the first group, already loaded; `sb` = LDS byte offset of the tile (a multiple
// of the tile size, so it commutes with the XOR addressing).  Ends with a barrier.
// The stage's ops are one contiguous stream (groups are emitted in order), so the scalar-load
// pipeline -- two descriptors and one matrix ahead of the gate being applied -- runs ACROSS group
// boundaries: a group's first gate never waits for descriptor -> matrix, and a group's 16 slot
// offsets are the XOR closure of four words fetched during the previous group.
// The 16 amplitudes are 16 scalars r0..r15 addressed with literal indices only: as an array
// walked by (unrolled) loops they become one <16 x i64> value early in the optimiser, a
// 32-register tuple that was then copied whole around every gate (32 v_mov_b64 per dense gate,
// a third of its instructions, until round 2's second profile pass found it).
// KEEP (whole-state launches): the 16 slot addresses of a group stay in VGPRs from the gather to the
// in-place scatter instead of being re-derived (16 v_xor per group); there the kernel has ~50 VGPRs
// to spare, the HBM-regime instantiations live on a 96-VGPR budget.  solo: the workgroup is ONE wave
// (10 qubits) -- its LDS operations execute in order, so a gather that follows a scatter needs no
// barrier and no drain of the LDS queue, only the compiler kept from reordering them.
template <bool KEEP>
__device__ __forceinline__ void tile2_groups(uint32_t sb, uint32_t addr, const Tile2Args &f,
                                             const u64 QMLE_CONSTANT *mrow, int tid, bool use_skip, bool solo) {
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  const Group2 QMLE_CONSTANT *grp = as_constant(f.groups);
  if (f.n_groups <= 0) return;
  // LoweredOp words: .y >> 24 = dispatch code, .z = matrix offset (floats)
  const v4u QMLE_CONSTANT *op = reinterpret_cast<const v4u QMLE_CONSTANT *>(as_constant(f.ops) + grp->op_begin);
  const int last = f.n_ops_stage > 0 ? f.n_ops_stage - 1 : 0;
  int k = 0;  // index into the stage's op stream
  v4u w0 = {0u, 0u, 0u, 0u}, w1 = {0u, 0u, 0u, 0u};
  Mat2S M0 = {0ull, 0ull, 0ull, 0ull};
  if (f.n_ops_stage > 0) {  // (a stage of layout changes only has no op and maybe no matrix row)
    w0 = op[0];
    w1 = op[last < 1 ? last : 1];
    const u64 QMLE_CONSTANT *m = mrow + (w0.z >> 1);
    M0 = {m[0], m[1], m[2], m[3]};
  }
  // header word (n_ops | relayout << 16) and the four basis offsets of the first group
  uint32_t hdr = reinterpret_cast<const uint32_t QMLE_CONSTANT *>(grp)[1];
  uint32_t o1 = grp->off[1], o2 = grp->off[2], o4 = grp->off[4], o8 = grp->off[8];
#define QMLE_OFF(c, b1, b2, b4, b8) \
  ((((c) & 1) ? (b1) : 0u) ^ (((c) & 2) ? (b2) : 0u) ^ (((c) & 4) ? (b4) : 0u) ^ (((c) & 8) ? (b8) : 0u))
  for (int gi = 0; gi < f.n_groups; ++gi, ++grp) {
    const int n_ops = (int)(hdr & 0xffffu);
    const bool relayout = ((hdr >> 16) & 0xffu) != 0;
    // bit 0 of the table entry: the work item's 16 amplitudes are known zeros at this point.
    // Wave-uniform use only: a wave of idle work items skips the group; an idle work item inside
    // a busy wave runs it on the zeros its slots hold (no per-lane branch around the gates)
    // (kept in an SGPR through readfirstlane: as a plain bool hipcc carried it across the gate loop as a lane mask
    // and re-materialised it with v_cndmask + v_cmp per GATE -- 2 of the 66 vector instructions of a dense gate)
    uint32_t busy_u = 1u;
    if (use_skip) busy_u = __builtin_amdgcn_ballot_w64(!(addr & 1u)) != 0ull ? 1u : 0u;
    int busy = __builtin_amdgcn_readfirstlane((int)busy_u);
    asm volatile("" : "+s"(busy));  // (an integer in an SGPR, compared where it is used: not an i1 lane mask)
    addr = (addr & ~7u) + sb;
    A16 r;
    uint32_t ka[16];  // KEEP: the slot addresses (indexed with literals only)
    if (busy != 0 || relayout) {  // (a relayout stores every slot of the new layout, zeros included)
      if (KEEP) {
#define QMLE_LD(c) ka[c] = addr ^ QMLE_OFF(c, o1, o2, o4, o8); r.v##c = lds_ld64(ka[c]);
        QMLE_X16(QMLE_LD)
#undef QMLE_LD
      } else {
#define QMLE_LD(c) r.v##c = lds_ld64(addr ^ QMLE_OFF(c, o1, o2, o4, o8));
        QMLE_X16(QMLE_LD)
#undef QMLE_LD
      }
    }
    // (!KEEP) the 16 slot addresses are re-derived for the scatter (16 v_xor) instead of living in 16
    // VGPRs across the gates: the kernel stays within 96 VGPRs = 5 waves per SIMD
    if (!KEEP) asm volatile("" : "+v"(addr));
    const bool more = gi + 1 < f.n_groups;
    uint32_t addr_next = 0;
    if (relayout) addr_next = f.tbl[grp->tbl_out + tid];
    else if (more) addr_next = f.tbl[grp[1].tbl + tid];
    // next group's header and basis offsets: in flight while this group's gates run
    const Group2 QMLE_CONSTANT *nx = more ? grp + 1 : grp;
    const uint32_t hdr_n = reinterpret_cast<const uint32_t QMLE_CONSTANT *>(nx)[1];
    const uint32_t n1 = nx->off[1], n2 = nx->off[2], n4 = nx->off[4], n8 = nx->off[8];
    for (int j = 0; j < n_ops; ++j, ++k) {
      // scalar loads return out of order, so only lgkmcnt(0) can cover them: touching this
      // gate's operands HERE puts that wait in front of the next prefetch instead of behind it
      asm volatile("" : "+s"(M0.m00), "+s"(M0.m01), "+s"(M0.m10), "+s"(M0.m11), "+s"(w0.y), "+s"(w1.z) :: "memory");
      if (f.dbg & 12) {  // timing anatomy only (wrong results): 4 = no per-gate scalar loads, 8 = + no dispatch
        if (busy != 0) {
          if (f.dbg & 8) f_dense<1>(r, M0);
          else fast_dispatch(r, (int)(w0.y >> 24), M0);
        }
        continue;
      }
      const u64 QMLE_CONSTANT *mn = mrow + (w1.z >> 1);
      const Mat2S Mn = {mn[0], mn[1], mn[2], mn[3]};
      const v4u w2 = op[k + 2 < last ? k + 2 : last];
      asm volatile("" : "+s"(busy));  // (re-read per gate: hoisted out of the loop the compare becomes a lane mask again)
      if (busy != 0) fast_dispatch(r, (int)(w0.y >> 24), M0);
      w0 = w1;
      w1 = w2;
      M0 = Mn;
    }
    if (relayout) {
      if (solo) asm volatile("" ::: "memory");
      else __syncthreads();  // every gather of the group is done: slots may change owners
      addr_next = (addr_next & ~7u) + sb;
      const uint32_t q1 = grp->off_out[1], q2 = grp->off_out[2], q4 = grp->off_out[4], q8 = grp->off_out[8];
#define QMLE_ST(c) lds_st64(addr_next ^ QMLE_OFF(c, q1, q2, q4, q8), r.v##c);
      QMLE_X16(QMLE_ST)
#undef QMLE_ST
      if (more) addr_next = f.tbl[grp[1].tbl + tid];
    } else if (busy != 0) {
      if (KEEP) {
#define QMLE_ST(c) lds_st64(ka[c], r.v##c);
        QMLE_X16(QMLE_ST)
#undef QMLE_ST
      } else {
#define QMLE_ST(c) lds_st64(addr ^ QMLE_OFF(c, o1, o2, o4, o8), r.v##c);
        QMLE_X16(QMLE_ST)
#undef QMLE_ST
      }
    }
    addr = addr_next;
    hdr = hdr_n;
    o1 = n1; o2 = n2; o4 = n4; o8 = n8;
    if (solo) asm volatile("" ::: "memory");
    else __syncthreads();
  }
#undef QMLE_OFF
}

// <Z> of every bit for the multi-tile measuring variant: the per-tile part only squares and adds
// (pruned Walsh-Hadamard butterfly over the four in-thread bits); the sums stay per work item
// across the tiles a workgroup walks -- acc = {total, h0..h3, total signed by bit 0 / 1 / 2 of
// the tile's index inside the walk} -- and the cross-lane reduction, the row assembly and the
// store run once per workgroup (`tile_z_finish`): no barrier, no DPP chain, no global store per
// tile.
__device__ __forceinline__ void tile_z_accumulate(uint32_t sbo, int T, int tid, int i, float (&acc)[8]) {
  uint32_t tid_e = (uint32_t)tid;
  asm volatile("" : "+v"(tid_e));  // (keeps the 16 addresses out of loop-carried registers)
  const uint32_t e0 = (sw(tid_e) << 3) + sbo;
  float pr[16];
#pragma unroll
  for (int h = 0; h < 16; h += 8) {
    u64 amp[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) amp[it] = lds_ld64(e0 ^ (sw((uint32_t)(h + it) << (T - 4)) << 3));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < 8; ++it)
      pr[h + it] = norm2(make_float2(__uint_as_float((uint32_t)amp[it]), __uint_as_float((uint32_t)(amp[it] >> 32))));
    __builtin_amdgcn_sched_barrier(0);
  }
  float h0 = 0.f, h1 = 0.f, h2 = 0.f, s1[8], s2[4], s3[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1[k] = pr[2 * k] + pr[2 * k + 1]; h0 += pr[2 * k] - pr[2 * k + 1]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { s2[k] = s1[2 * k] + s1[2 * k + 1]; h1 += s1[2 * k] - s1[2 * k + 1]; }
#pragma unroll
  for (int k = 0; k < 2; ++k) { s3[k] = s2[2 * k] + s2[2 * k + 1]; h2 += s2[2 * k] - s2[2 * k + 1]; }
  const float tot = s3[0] + s3[1];
  acc[0] += tot;
  acc[1] += h0;
  acc[2] += h1;
  acc[3] += h2;
  acc[4] += s3[0] - s3[1];
  acc[5] += (i & 1) ? -tot : tot;  // (i is wave-uniform)
  acc[6] += (i & 2) ? -tot : tot;
  acc[7] += (i & 4) ? -tot : tot;
}
// Row [33] of workgroup `row` of sample b: thread q < n assembles <Z> of position q from qsrc
// (see TileArgs::qsrc): lane bit, in-thread bit, wave bit, or outer position = bit of the tile
// index -- one of the `lg` walk bits (own signed sums) or a bit of the workgroup index.
__device__ __forceinline__ void tile_z_finish(float *out, float *red, float (&acc)[8], int qsrc, int tid,
                                              int nt, int lg, uint32_t row, uint32_t n_rows, int b) {
  const int lane = tid & (kWave - 1), w = tid / kWave, nw = (nt + kWave - 1) / kWave;
  float v[14];
#pragma unroll
  for (int j = 0; j < 6; ++j) v[j] = ((lane >> j) & 1) ? -acc[0] : acc[0];
  v[6] = acc[1]; v[7] = acc[2]; v[8] = acc[3]; v[9] = acc[4]; v[10] = acc[0];
  v[11] = acc[5]; v[12] = acc[6]; v[13] = acc[7];
  wave_sums_dpp63(v);
  __syncthreads();  // every amplitude of the last tile has been read: the tile buffer is scratch
  if (lane == kWave - 1) {
#pragma unroll
    for (int j = 0; j < 14; ++j) red[w * 14 + j] = v[j];
  }
  __syncthreads();
  if (tid <= QMLE_MAX_QUBITS) {
    float r = 0.f;
    if (qsrc < 16) {
      for (int k = 0; k < nw; ++k) r += red[k * 14 + qsrc];
    } else if (qsrc < 32) {
      for (int k = 0; k < nw; ++k) r += ((k >> (qsrc - 16)) & 1) ? -red[k * 14 + 10] : red[k * 14 + 10];
    } else if (qsrc < 64) {
      const int t = qsrc - 32;
      if (t < lg) {
        for (int k = 0; k < nw; ++k) r += red[k * 14 + 11 + t];
      } else {
        for (int k = 0; k < nw; ++k) r += red[k * 14 + 10];
        if ((row >> (t - lg)) & 1u) r = -r;
      }
    }
    out[((size_t)b * n_rows + row) * (QMLE_MAX_QUBITS + 1) + tid] = r;
  }
}

// Z-parity observables (TM_EXPVAL_MASKS) for the multi-tile measuring variant (round 5).  Folded CX tails turn <Z_w>
// into parities, and until now such a last pass kept one tile per workgroup with the full epilogue (two barriers, a
// round of wave sums, a row) per tile: 58 us per state for the three groups of the deep default plan at n = 24, where
// the single-bit form above takes 23 for the same work.  Per tile the work item squares its 16 amplitudes, runs the
// 16-point Walsh-Hadamard butterfly over the iteration bits and adds, per observable, the ONE sum it needs (a
// wave-uniform pick) -- signed by the parity of this tile's index under the observable's outer bits (wave-uniform) --
// into the observable's own accumulator; lane signs, wave sums, wave-index signs and the row come once per walk.
constexpr int kMaskMultiObs = 24;  // accumulators per work item (observables of a launch; more: one tile per workgroup)
__device__ __forceinline__ float pick16(const float (&w)[16], uint32_t i) {  // i is wave-uniform
  switch (i) {
    case 0: return w[0]; case 1: return w[1]; case 2: return w[2]; case 3: return w[3];
    case 4: return w[4]; case 5: return w[5]; case 6: return w[6]; case 7: return w[7];
    case 8: return w[8]; case 9: return w[9]; case 10: return w[10]; case 11: return w[11];
    case 12: return w[12]; case 13: return w[13]; case 14: return w[14]; default: return w[15];
  }
}
__device__ __forceinline__ void tile_m_accumulate(uint32_t sbo, int T, int tid, uint32_t tile, int n_obs,
                                                  const uint16_t QMLE_CONSTANT *ol, const uint32_t QMLE_CONSTANT *oo,
                                                  float (&acc)[kMaskMultiObs]) {
  uint32_t tid_e = (uint32_t)tid;
  asm volatile("" : "+v"(tid_e));  // (keeps the 16 addresses out of loop-carried registers)
  const uint32_t e0 = (sw(tid_e) << 3) + sbo;
  float w[16];
#pragma unroll
  for (int h = 0; h < 16; h += 8) {
    u64 amp[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) amp[it] = lds_ld64(e0 ^ (sw((uint32_t)(h + it) << (T - 4)) << 3));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < 8; ++it)
      w[h + it] = norm2(make_float2(__uint_as_float((uint32_t)amp[it]), __uint_as_float((uint32_t)(amp[it] >> 32))));
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int h = 1; h < 16; h <<= 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i & h) continue;
      const float x = w[i], y = w[i | h];
      w[i] = x + y;
      w[i | h] = x - y;
    }
  }
  static_for<kMaskMultiObs>([&](auto k) {
    if ((int)k < n_obs) {  // (wave-uniform)
      const float sel = pick16(w, (uint32_t)ol[(int)k] >> (T - 4));
      const uint32_t flip = (uint32_t)(__builtin_popcount(tile & oo[(int)k]) & 1) << 31;
      acc[k] += __uint_as_float(__float_as_uint(sel) ^ flip);
    }
  });
}
// one row [QMLE_MAX_QUBITS + 1] per workgroup: column k = observable k (like the one-tile epilogue's rows)
__device__ __forceinline__ void tile_m_finish(float *out, float *red, const float (&acc)[kMaskMultiObs], int n_obs,
                                              const uint16_t QMLE_CONSTANT *ol, int T, int tid, int nt, uint32_t row,
                                              uint32_t n_rows, int b) {
  const int lane = tid & (kWave - 1), wv = tid / kWave, nw = (nt + kWave - 1) / kWave;
  __syncthreads();  // every amplitude of the last tile has been read: the tile buffer is scratch
  static_for<kMaskMultiObs / 8>([&](auto r8) {
    if (8 * (int)r8 < n_obs) {
      float v[8];
      static_for<8>([&](auto k) {
        constexpr int kk = 8 * (int)r8 + (int)k;
        const uint32_t ml = kk < n_obs ? ((uint32_t)ol[kk] & 63u) : 0u;
        v[k] = (__popc((uint32_t)lane & ml) & 1) ? -acc[kk] : acc[kk];
      });
      wave_sums_dpp63(v);
      if (lane == kWave - 1) static_for<8>([&](auto k) { red[(8 * (int)r8 + (int)k) * nw + wv] = v[k]; });
    }
  });
  __syncthreads();
  if (tid < n_obs) {
    const uint32_t mw = ((uint32_t)ol[tid] >> 6) & ((1u << (T - 10)) - 1u);
    float r = 0.f;
    for (int v = 0; v < nw; ++v) {
      const float c = red[tid * nw + v];
      r += (__popc((uint32_t)v & mw) & 1) ? -c : c;
    }
    out[((size_t)b * n_rows + row) * (QMLE_MAX_QUBITS + 1) + tid] = r;
  }
}

// TM_EXPVAL of k_tile2's whole-state tile (T == n >= 10: the tile index IS the amplitude index).
// Every work item squares its 16 amplitudes once, in the load stage's layout: slot 2u + e has
// index bit 0 = e and bits T-3.. = u, the lane holds bits 1..6, the wave index bits 7..T-4.  The
// sign of an observable splits accordingly: the slot part is wave-uniform (8 signed adds on
// sums or differences of slot pairs), the lane part one popcount, the wave part is applied by the
// final sum.  Eight observables per round of DPP wave sums, one barrier in all -- tile_epilogue's
// loop re-read the tile and ran a block sum per observable (a quarter of the kernel at 10 qubits).
__device__ __forceinline__ void whole_state_expval(const TileArgs &a, uint32_t sl, const uint32_t (&soff)[8],
                                                   float *red, int tid, int nt, int b) {
  const int T = a.T, lane = tid & (kWave - 1), wv = tid / kWave;
  const int nw = nt >= kWave ? nt / kWave : 1;  // <= 8
  float S[8], Df[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const float4 w = lds_ld128(sl ^ soff[u]);
    const float p0 = w.x * w.x + w.y * w.y, p1 = w.z * w.z + w.w * w.w;
    S[u] = p0 + p1;
    Df[u] = p0 - p1;
  }
  // (through the kernel argument segment: indexing the by-value struct with a run-time index
  // makes hipcc copy it to scratch)
  const uint32_t QMLE_CONSTANT *om =
      (const uint32_t QMLE_CONSTANT *)((const char QMLE_CONSTANT *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TileArgs, obs_mask));
  const int n_obs = a.n_obs;
  for (int k0 = 0; k0 < n_obs; k0 += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = 0.f;
      if (k0 + k < n_obs) {  // (wave-uniform)
        const uint32_t m = om[k0 + k];
        const uint32_t mu = (m >> (T - 3)) & 7u;
        float t = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float d = (m & 1u) ? Df[u] : S[u];
          t += (__popc((uint32_t)u & mu) & 1) ? -d : d;
        }
        v[k] = (__popc((uint32_t)lane & (m >> 1) & 63u) & 1) ? -t : t;
      }
    }
    wave_sums_dpp63(v);
    if (lane == kWave - 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) red[(k0 + k) * nw + wv] = v[k];  // (k0 + k < 32: launch_tile sizes red as 32 x nw)
    }
  }
  __syncthreads();
  if (tid < n_obs) {
    const uint32_t mw = om[tid] >> 7;
    float r = 0.f;
    for (int w = 0; w < nw; ++w) r += (__popc((uint32_t)w & mw) & 1) ? -red[tid * nw + w] : red[tid * nw + w];
    reinterpret_cast<float *>(a.out)[(size_t)b * n_obs + tid] = r;
  }
}

// MEASURE: a.meas is one of the TM_EXPVAL_* epilogues (own instantiation: the storing kernel keeps
// a small register budget).  MULTI: several tiles per workgroup (f.tpw), plain all-live stages
// with the TM_STORE / TM_PROBS / TM_EXPVAL_PARTIAL epilogues only.
// WS: the tile is the whole state (T == n, 10..13 qubits, launch_tile): the group loop keeps its slot
// addresses in registers and a one-wave workgroup drops its barriers (tile2_groups).
// MW (with MEASURE, without MULTI): a.meas is TM_STORE_MW / TM_MW_ONLY -- the tile's Meyer-Wallach
// row (tile_mw_row) behind the store; its own instantiations, so that the 40 sums it keeps per work item
// do not enter the register budget of the others.
// MASKS (with MEASURE and MULTI): a.meas is TM_EXPVAL_MASKS, <= kMaskMultiObs observables (tile_m_accumulate / _finish).
template <bool NT, bool MEASURE, bool MULTI, bool WS = false, bool MW = false, bool MASKS = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(MASKS ? 4 : 5))) k_tile2(const TileArgs a, const Tile2Args f) {
  extern __shared__ float4 smem4[];
  float2 *s = reinterpret_cast<float2 *>(smem4);
  const int T = a.T;
  // LDS byte offset of the tile; the XOR addressing needs it aligned to the tile size (it is 0:
  // the kernel has no static LDS)
  // (launch_tile checks this on the host, once per device: lds_base_is_zero)
  const uint32_t sbo = lds_offset_of(smem4);
  // measuring epilogues: scratch aliases the tile, except the whole-state one (own region)
  float *red = a.meas == TM_EXPVAL ? reinterpret_cast<float *>(s + (1u << T)) : reinterpret_cast<float *>(s);
  const int tid = threadIdx.x, nt = blockDim.x;  // nt = 2^(T-4)
  const int b = blockIdx.y;
  // Plain all-live stages give a workgroup `tpw` consecutive tiles (MULTI).  Storing passes keep
  // the next tile's 8 float4 per lane in flight in registers while this tile's gates run;
  // measuring passes walk without that prefetch but keep their <Z> sums in registers across the
  // walk and reduce once (tile_z_accumulate / tile_z_finish).  Both save the workgroup turnover
  // (launch gap + prologue) per tile.  K2 at n = 24: read+write pass 54 -> 51 us per state,
  // measuring pass 29 -> 23.6.  Known-zero stages keep one tile per workgroup.
  const int tpw = MULTI ? f.tpw : 1;
  uint32_t tile = blockIdx.x * (uint32_t)tpw;
  if (!MULTI && a.compact) {  // blockIdx.x enumerates the tiles that can be non-zero (launch_tile)
    uint32_t rest = tile, free_bits = a.tile_free;
    tile = 0;
    while (rest) {
      const uint32_t low = free_bits & (0u - free_bits);
      if (rest & 1u) tile |= low;
      free_bits ^= low;
      rest >>= 1;
    }
  }
  const uint32_t n_tiles = gridDim.x * (uint32_t)tpw;
  const size_t D = (size_t)1 << a.n;
  uint64_t base = tile2_base(a, f, tile);
  // global addresses: wave-uniform 64-bit base (SGPRs) + one 32-bit byte offset per lane
  char *st = reinterpret_cast<char *>(a.states + (size_t)b * D + base);
  // a lane's 8 float4: local index j = 2 (tid + u nt): bit 0 rides in the access, bits 1..T-4
  // come from tid, the top three from u (wave-uniform)
  // (both come precomputed: indexing the int8 position arrays of the kernel arguments with
  // run-time indices costs a chain of vector loads in front of the tile's own loads)
  const uint32_t jl = 2u * tid;
  uint32_t goff8;  // < 2^31 for n <= 28
  if (f.n_in_runs < 0) {
    goff8 = f.tbl[f.gtab + tid];
  } else {
    uint32_t g = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r < f.n_in_runs) g |= ((jl >> f.in_off[r]) & f.in_mask[r]) << f.in_pos[r];
    goff8 = g << 3;
  }
  uint32_t uoff[8], soff[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    uoff[u] = f.uoff8[u];
    soff[u] = sw((uint32_t)u << (T - 3)) << 3;  // LDS byte offset; sw() is linear over XOR
  }
  const uint32_t sl = (sw(jl) << 3) + sbo;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!MULTI && (a.init_zero ? base != 0 : (tile & a.zin_outer) != 0)) {
    // |0..0> lives in tile 0 alone and gates are linear: a tile that holds only known zeros
    // (Stage::zero_in) stays exactly zero -- write the zeros, skip the gates
    if (MW) {  // a zero tile contributes a zero row (and, TM_STORE_MW, its zeros to the state)
      if (a.meas == TM_STORE_MW) {
#pragma unroll
        for (int u = 0; u < 8; ++u) st4<NT>(reinterpret_cast<float4 *>(st + uoff[u] + goff8), z4);
      }
      float *po = reinterpret_cast<float *>(a.out) + ((size_t)b * n_tiles + tile) * kMwFusedRow;
      if (tid < kMwFusedRow) po[tid] = 0.f;
    } else if (!MEASURE && a.meas == TM_STORE) {
#pragma unroll
      for (int u = 0; u < 8; ++u) st4<NT>(reinterpret_cast<float4 *>(st + uoff[u] + goff8), z4);
    } else if (!MEASURE) {
      char *po = reinterpret_cast<char *>(reinterpret_cast<float *>(a.out) + (size_t)b * D + base);
#pragma unroll
      for (int u = 0; u < 8; ++u) *reinterpret_cast<float2 *>(po + (uoff[u] >> 1) + (goff8 >> 1)) = make_float2(0.f, 0.f);
    } else {
      float *po = reinterpret_cast<float *>(a.out) +
                  ((size_t)b * n_tiles + tile) * (QMLE_MAX_QUBITS + 1);
      if (tid <= QMLE_MAX_QUBITS) po[tid] = 0.f;
    }
    return;
  }
  const Group2 QMLE_CONSTANT *grp = as_constant(f.groups);
  const uint32_t addr = f.n_groups > 0 ? f.tbl[grp->tbl + tid] : 0u;  // in flight beside the tile
  const u64 QMLE_CONSTANT *mrow = as_constant(reinterpret_cast<const u64 *>(a.mats + (size_t)b * a.mat_floats));
  // (TM_EXPVAL_PARTIAL: where thread q finds <Z> of position q -- read once, through the kernel
  // argument segment: indexing the by-value struct inside the tile loop makes hipcc copy it to
  // scratch)
  int qsrc = -1;
  if (MEASURE) {
    const int8_t QMLE_CONSTANT *ka = (const int8_t QMLE_CONSTANT *)__builtin_amdgcn_kernarg_segment_ptr();
    qsrc = tid <= QMLE_MAX_QUBITS ? (int)ka[offsetof(TileArgs, qsrc) + tid] : 64;
  }
  const bool plain = MULTI || (!a.init_zero && !a.zin_local);
  float4 v[8];
  // the 8 loads of a tile.  Multi-tile walks over a stage with known zeros INSIDE the tile
  // (zin_local; every tile live: launch_tile) read only the amplitudes that can be non-zero, like
  // the one-tile path below -- round 3: those stages used to keep one tile per workgroup, and at
  // T = 13 (two workgroups per CU, started together and finishing together) their loads and their
  // gate groups never overlapped: 34 us of traffic + 80 us of arithmetic = 113 us for the 9-group
  // pass of the default engine's deep run (profiles/r03_deep_default_anatomy.txt)
  const uint32_t zl_m = MULTI ? (a.zin_local & ~1u) : 0u;
  const bool z0_m = MULTI && (a.zin_local & 1u) != 0;
  auto load_tile = [&](const char *p) {
    if (f.dbg & 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = make_float4(1e-3f, 0.f, 1e-3f, 0.f);
    } else if (MULTI && a.zin_local) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = z4;
        if (((jl | ((uint32_t)u << (T - 3))) & zl_m) == 0) {
          v[u] = ld4<NT>(reinterpret_cast<const float4 *>(p + uoff[u] + goff8));
          if (z0_m) v[u].z = v[u].w = 0.f;
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ld4<NT>(reinterpret_cast<const float4 *>(p + uoff[u] + goff8));
    }
  };
  if (plain) load_tile(st);
  const uint32_t sl_outer = sl;
  float zacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // MEASURE && MULTI: tile_z_accumulate
  float pacc[MASKS ? kMaskMultiObs : 1];                      // MASKS: tile_m_accumulate
  if (MASKS) static_for<kMaskMultiObs>([&](auto k) { pacc[MASKS ? (int)k : 0] = 0.f; });
  const uint16_t QMLE_CONSTANT *m_ol =
      (const uint16_t QMLE_CONSTANT *)((const char QMLE_CONSTANT *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TileArgs, obs_local));
  const uint32_t QMLE_CONSTANT *m_oo =
      (const uint32_t QMLE_CONSTANT *)((const char QMLE_CONSTANT *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TileArgs, obs_outer));
  MwAcc macc;                                                 // MW: tile_mw_accumulate
  if (MW) tile_mw_clear(macc);
  for (int i = 0; i < tpw; ++i) {
    uint32_t sl = sl_outer;  // (opaque per tile: keeps the 8 staging addresses out of loop-carried registers)
    if (MULTI) asm volatile("" : "+v"(sl));
    if (MULTI && MEASURE && i > 0) {
      // measuring passes walk their tiles without prefetch (measured with it, before and after
      // the sums moved into registers: no gain)
      base += f.tile_stride;
      st += f.tile_stride * sizeof(float2);
      load_tile(st);
    }
    if (!MULTI && a.init_zero) {
#pragma unroll
      for (int u = 0; u < 8; ++u) lds_st128(sl ^ soff[u], z4);
      __syncthreads();
      if (tid == 0) s[sw(0)] = make_float2(1.f, 0.f);  // |0...0>, simulation.py:100
    } else if (!MULTI && a.zin_local) {
      // only the amplitudes that can be non-zero are read; the rest of the tile is zero-filled
      const uint32_t zl = a.zin_local & ~1u;
      const bool z0 = (a.zin_local & 1u) != 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = z4;
        if (((jl | ((uint32_t)u << (T - 3))) & zl) == 0) {
          v[u] = ld4<NT>(reinterpret_cast<const float4 *>(st + uoff[u] + goff8));
          if (z0) v[u].z = v[u].w = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) lds_st128(sl ^ soff[u], v[u]);
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) lds_st128(sl ^ soff[u], v[u]);
    }
    __syncthreads();
    char *st_cur = st;
    const uint64_t base_cur = base;
    if (!MEASURE && i + 1 < tpw) {  // (plain storing stages only) the next tile: loads in flight from here on
      base += f.tile_stride;  // (the tiles of a walk differ in the lowest outer bits only: launch_tile)
      st += f.tile_stride * sizeof(float2);
      load_tile(st);
    }

    tile2_groups<WS>(sbo, addr, f, mrow, tid, a.zin_local != 0, WS && nt <= kWave);  // known zeros: Stage::zero_in

    if (MW) {
      if (a.meas == TM_STORE_MW) {
#pragma unroll
        for (int h = 0; h < 8; h += 4) {  // (two batches of four: the sums of the walk are live)
          float4 w[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) w[u] = lds_ld128(sl ^ soff[h + u]);
#pragma unroll
          for (int u = 0; u < 4; ++u) st4<NT>(reinterpret_cast<float4 *>(st_cur + uoff[h + u] + goff8), w[u]);
        }
      }
      tile_mw_accumulate(sbo, T, (uint32_t)tid, (uint32_t)i, macc, a.mw_lean != 0);
    } else if (MEASURE && MULTI && MASKS) {  // (TM_EXPVAL_MASKS: launch_tile)
      if constexpr (MASKS) tile_m_accumulate(sbo, T, tid, tile + (uint32_t)i, a.n_obs, m_ol, m_oo, pacc);
    } else if (MEASURE && MULTI) {  // (TM_EXPVAL_PARTIAL only: launch_tile)
      if (!(f.dbg & 2)) tile_z_accumulate(sbo, T, tid, i, zacc);
    } else if (MEASURE) {
      if (!(f.dbg & 2)) {
        if (a.meas == TM_EXPVAL) whole_state_expval(a, sl, soff, red, tid, nt, b);
        else tile_epilogue<false>(a, s, nullptr, red, tile + (uint32_t)i, n_tiles, b, base_cur, qsrc);
      }
    } else if (a.meas == TM_STORE) {
      if (MULTI) {  // (the next tile's 8 float4 are live: two batches of four keep <= 96 VGPRs)
#pragma unroll
        for (int h = 0; h < 8; h += 4) {
          float4 w[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) w[u] = lds_ld128(sl ^ soff[h + u]);
#pragma unroll
          for (int u = 0; u < 4; ++u) if (!(f.dbg & 16) || w[u].x == 123.f) st4<NT>(reinterpret_cast<float4 *>(st_cur + uoff[h + u] + goff8), w[u]);
        }
      } else {
        float4 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = lds_ld128(sl ^ soff[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) if (!(f.dbg & 16) || w[u].x == 123.f) st4<NT>(reinterpret_cast<float4 *>(st_cur + uoff[u] + goff8), w[u]);
      }
    } else {
      char *po = reinterpret_cast<char *>(reinterpret_cast<float *>(a.out) + (size_t)b * D + base_cur);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 w = lds_ld128(sl ^ soff[u]);
        *reinterpret_cast<float2 *>(po + (uoff[u] >> 1) + (goff8 >> 1)) =
            make_float2(w.x * w.x + w.y * w.y, w.z * w.z + w.w * w.w);
      }
    }
    if (i + 1 < tpw) __syncthreads();  // the tile buffer (and the epilogue's scratch in it) is reused
  }
  if (MW)  // one row per workgroup (= per tile: launch_tile keeps these stages at one tile per workgroup)
    tile_mw_finish(macc, T, (uint32_t)tid, reinterpret_cast<float *>(s),
                   reinterpret_cast<float *>(a.out) + ((size_t)b * gridDim.x + blockIdx.x) * kMwFusedRow, a.mw_lean != 0);
  else if (MEASURE && MULTI && MASKS) {
    if constexpr (MASKS)
      tile_m_finish(reinterpret_cast<float *>(a.out), red, pacc, a.n_obs, m_ol, T, tid, nt, blockIdx.x, gridDim.x, b);
  } else if (MEASURE && MULTI && !(f.dbg & 2))
    tile_z_finish(reinterpret_cast<float *>(a.out), red, zacc, qsrc, tid, nt, 31 - __builtin_clz((unsigned)tpw),
                  blockIdx.x, gridDim.x, b);
}

// ---- measuring pass in registers -----------------------------------------------------------
// Last pass of a <Z> / Z-parity run whose gates all sit on <= 4 bit positions (ONE register-tile
// group): nothing is staged through LDS.  Every work item loads its 16 amplitudes straight from
// HBM (only those that can be non-zero, TileArgs::zin_local), applies the gates in registers and
// turns |a|^2 into the 16 Walsh-Hadamard sums over its 4 bits.  A workgroup walks 2^q tiles of
// one state; the observable's sum over tiles is accumulated PER WORK ITEM (sign = parity of
// the tile index under the observable's outer bits, one bit mask S for all observables,
// updated with one XOR per tile), so the cross-lane signed reduction runs once per workgroup
// instead of once per tile.  Row layout as TM_EXPVAL_MASKS: out[b][workgroup][k < n_obs].

template <bool FOLD>
__global__ void __launch_bounds__(1024) k_reg_measure(const TileArgs a, int q) {
  extern __shared__ float4 smem4[];
  OpSlot *slots = reinterpret_cast<OpSlot *>(smem4);
  uint32_t *meta = reinterpret_cast<uint32_t *>(slots + a.n_ops);
  uint32_t *m_thr = meta;        // [32] observable restricted to the work-item bits
  uint32_t *m_reg = meta + 32;   // [32] ... to the 4 register bits
  uint32_t *m_pack = meta + 64;  // [4]  the same, 8 x 4 bits per word
  uint32_t *flipF = meta + 68;   // [32] bit k: observable k contains outer bit j
  uint32_t *flipP = meta + 100;  // [32] prefix XOR of flipF
  float *red = reinterpret_cast<float *>(meta + 132);  // [16][32]
  const int T = a.T, tid = threadIdx.x, b = blockIdx.y;
  const int n_outer = a.n - T;
  const OpGroup g = a.groups[0];
  const int b0 = g.bits[0], b1 = g.bits[1], b2 = g.bits[2], b3 = g.bits[3];

  tile_stage_slots(a, slots, b);
  if (tid < 32) {
    uint32_t mt = 0, mi = 0;
    if (tid < a.n_obs) {
      const uint32_t m = a.obs_mask[tid];
      int tb = 0;
      for (int j = 0; j < T; ++j) {
        const uint32_t bitv = (m >> a.tile_bits[j]) & 1u;
        if (j == b0) mi |= bitv;
        else if (j == b1) mi |= bitv << 1;
        else if (j == b2) mi |= bitv << 2;
        else if (j == b3) mi |= bitv << 3;
        else mt |= bitv << tb++;
      }
    }
    m_thr[tid] = mt;
    m_reg[tid] = mi;
    uint32_t f = 0;
    if (tid < n_outer)
      for (int k = 0; k < a.n_obs; ++k) f |= ((a.obs_mask[k] >> a.outer_bits[tid]) & 1u) << k;
    flipF[tid] = f;
  }
  __syncthreads();
  if (tid < 32) {
    uint32_t pre = 0;
    for (int j = 0; j <= tid; ++j) pre ^= flipF[j];
    flipP[tid] = pre;
    if (tid < 4) {
      uint32_t w = 0;
      for (int k = 0; k < 8; ++k) w |= m_reg[tid * 8 + k] << (4 * k);
      m_pack[tid] = w;
    }
  }
  __syncthreads();
  const uint32_t mi0 = __builtin_amdgcn_readfirstlane(m_pack[0]);
  const uint32_t mi1 = __builtin_amdgcn_readfirstlane(m_pack[1]);
  const uint32_t mi2 = __builtin_amdgcn_readfirstlane(m_pack[2]);
  const uint32_t mi3 = __builtin_amdgcn_readfirstlane(m_pack[3]);

  // this work item's 16 amplitudes: local index lb | off(c), element offset gbase + goff(c)
  const uint32_t lb = ins0(ins0(ins0(ins0((uint32_t)tid, b0), b1), b2), b3);
  uint32_t gbase = 0;
  for (int j = 0; j < T; ++j) gbase |= ((lb >> j) & 1u) << a.tile_bits[j];
  const bool thread_ok = (lb & a.zin_local) == 0;
  const uint32_t G0 = 1u << a.tile_bits[b0], G1 = 1u << a.tile_bits[b1];
  const uint32_t G2 = 1u << a.tile_bits[b2], G3 = 1u << a.tile_bits[b3];
  uint32_t c_ok = 0;  // register slots that can be non-zero
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const uint32_t offl = ((c & 1) ? (1u << b0) : 0u) | ((c & 2) ? (1u << b1) : 0u) |
                          ((c & 4) ? (1u << b2) : 0u) | ((c & 8) ? (1u << b3) : 0u);
    if ((offl & a.zin_local) == 0) c_ok |= 1u << c;
  }
  const float2 *st = a.states + ((size_t)b << a.n);

  const uint32_t tile0 = blockIdx.x << q;
  uint32_t S = 0;  // bit k: sign of observable k on the current tile
  for (int j = 0; j < n_outer; ++j)
    if ((tile0 >> j) & 1u) S ^= flipF[j];
  S = __builtin_amdgcn_readfirstlane(S);

  // FOLD: few live inputs (<= 4 of the 16 register slots can be non-zero) -- the group's gates
  // act on known zeros almost everywhere, so the work item's 16 outputs are sum_j in_j * (U e_j)
  // and the columns U e_j are the same for the whole workgroup.  The gate code runs once, on
  // the basis vectors, and parks the columns in LDS; a tile then costs `cols` complex
  // multiply-adds per amplitude instead of the whole gate list.
  const int cols = __popc(c_ok);
  float2 *tcol = reinterpret_cast<float2 *>(red + 16 * 32);  // [4][16]
  uint32_t ingo[4];  // element offset of live input j
  {
    int okc[4];  // register slot of live input j
    uint32_t rest = c_ok;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      okc[j] = rest ? __builtin_ctz(rest) : 0;
      rest &= rest - 1u;
      ingo[j] = ((okc[j] & 1) ? G0 : 0u) | ((okc[j] & 2) ? G1 : 0u) | ((okc[j] & 4) ? G2 : 0u) |
                ((okc[j] & 8) ? G3 : 0u);
    }
    if (FOLD) {
      float2 v[16];
      const int mine = okc[(tid & 3) < cols ? (tid & 3) : 0];
#pragma unroll
      for (int c = 0; c < 16; ++c) v[c] = make_float2(c == mine ? 1.f : 0.f, 0.f);
      reg_apply_group(v, g, slots, a.op_begin);
      if (tid < cols) {
#pragma unroll
        for (int c = 0; c < 16; ++c) tcol[tid * 16 + c] = v[c];
      }
      __syncthreads();
    }
  }

  float A[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) A[k] = 0.f;

  // FOLD: the (few) inputs of the next tile are requested before this tile is worked on
  float2 nxt[4];
  auto fetch = [&](uint32_t tile) {
    const float2 *pt = st + tile_base(a, tile) + gbase;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      nxt[j] = make_float2(0.f, 0.f);
      if (j < cols && thread_ok && (tile & a.zin_outer) == 0) nxt[j] = pt[ingo[j]];
    }
  };
  if (FOLD) fetch(tile0);

  for (uint32_t it = 0; it < (1u << q); ++it) {
    const uint32_t tile = tile0 + it;
    float2 in[4];
    if (FOLD) {
#pragma unroll
      for (int j = 0; j < 4; ++j) in[j] = nxt[j];
      if (it + 1 < (1u << q)) fetch(tile + 1);
    }
    if ((tile & a.zin_outer) == 0) {
      const float2 *pt = st + tile_base(a, tile) + gbase;
      float2 v[16];
      if (FOLD) {
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = cmul(tcol[c], in[0]);
        for (int j = 1; j < cols; ++j) {
          const float2 x = j == 1 ? in[1] : j == 2 ? in[2] : in[3];
#pragma unroll
          for (int c = 0; c < 16; ++c) v[c] = cfma(tcol[j * 16 + c], x, v[c]);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const uint32_t go = ((c & 1) ? G0 : 0u) | ((c & 2) ? G1 : 0u) | ((c & 4) ? G2 : 0u) |
                              ((c & 8) ? G3 : 0u);
          v[c] = make_float2(0.f, 0.f);
          if (((c_ok >> c) & 1u) && thread_ok) v[c] = pt[go];
        }
        reg_apply_group(v, g, slots, a.op_begin);
      }
      v16f W;
#pragma unroll
      for (int c = 0; c < 16; ++c) W[c] = norm2(v[c]);
#pragma unroll
      for (int h = 1; h < 16; h <<= 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i & h) continue;
          const float x = W[i], y = W[i | h];
          W[i] = x + y;
          W[i | h] = x - y;
        }
      }
#pragma unroll
      for (int k0 = 0; k0 < 32; k0 += 8) {
        if (k0 < a.n_obs) {
          const uint32_t pack = k0 == 0 ? mi0 : k0 == 8 ? mi1 : k0 == 16 ? mi2 : mi3;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float w = W[(pack >> (4 * k)) & 15u];  // wave-uniform register index
            A[k0 + k] += ((S >> (k0 + k)) & 1u) ? -w : w;
          }
        }
      }
    }
    // next tile: the bits 0 .. (trailing ones of it) of the tile index flip
    const int tz = __builtin_ctz(~it);
    S ^= __builtin_amdgcn_readfirstlane(flipP[tz < 31 ? tz : 31]);
  }

  // signed sum over the work items: parity of the work-item index under the observable
#pragma unroll
  for (int k = 0; k < 32; ++k)
    if (__popc((uint32_t)tid & m_thr[k]) & 1) A[k] = -A[k];
  const int lane = tid & (kWave - 1), wv = tid / kWave, nw = (blockDim.x + kWave - 1) / kWave;
  const float mine = wave_reduce_scatter<32>(A);
  if (lane < 32) red[wv * 32 + lane] = mine;
  __syncthreads();
  float *po = reinterpret_cast<float *>(a.out) +
              ((size_t)b * gridDim.x + blockIdx.x) * (QMLE_MAX_QUBITS + 1);
  if (tid <= QMLE_MAX_QUBITS) {
    float r = 0.f;
    if (tid < a.n_obs)
      for (int w = 0; w < nw; ++w) r += red[w * 32 + tid];
    po[tid] = r;
  }
}

// k_reg_measure when ALL FOUR register bits are known-zero on input: the work item reads one
// amplitude x per tile and its 16 outputs are x * (U e_0), so every Walsh-Hadamard sum of the
// tile is |x|^2 times a number that is the same for the whole workgroup (coef_k, from U e_0).
// What is left per tile is |x|^2; the signs of the 32 tiles a workgroup walks (parity of the
// tile index under the observable's outer bits) are applied by ONE more Walsh-Hadamard
// transform, over the tile axis, held in registers: 32 independent 8-byte loads in flight per
// work item, 2.5 adds per tile, and one signed cross-lane reduction per workgroup.
typedef float v32f __attribute__((ext_vector_type(32)));

// Per observable: its wires split by where the measuring pass finds them (host-computed).
struct MonoObs {
  uint32_t thr[32];  // ... among the work-item bits (tile bits outside the gate group)
  uint32_t out[32];  // ... among the outer (tile index) bits
  uint8_t reg[32];   // ... among the 4 register bits
};

// coef[b][k] = Walsh-Hadamard sum `reg[k]` of |U e_0|^2 for sample b's gate group: one work
// item per sample (the gate list runs once per state instead of once per workgroup).
__global__ void __launch_bounds__(64)
k_mono_coef(const LoweredOp *__restrict__ ops, const OpGroup *__restrict__ group,
            const float *__restrict__ mats, uint32_t mat_floats, const MonoObs mo, int n_obs,
            float *__restrict__ coef, int batch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const OpGroup g = group[0];
  const float *mrow = mats + (size_t)b * mat_floats;
  float2 v[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) v[c] = make_float2(c == 0 ? 1.f : 0.f, 0.f);
  for (int k = 0; k < g.n_ops; ++k) {
    const LoweredOp op = ops[g.op_begin + k];
    const Mat2 m = load_mat2(mrow + op.mat_off);
    const int cb = op.nc ? op.c0 : -1;
    if (op.flags & LF_PERMX) reg_dispatch<2>(v, m, cb, op.t0);
    else if (op.flags & LF_DIAG) reg_dispatch<1>(v, m, cb, op.t0);
    else reg_dispatch<0>(v, m, cb, op.t0);
  }
  v16f W;
#pragma unroll
  for (int c = 0; c < 16; ++c) W[c] = norm2(v[c]);
#pragma unroll
  for (int h = 1; h < 16; h <<= 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i & h) continue;
      const float x = W[i], y = W[i | h];
      W[i] = x + y;
      W[i | h] = x - y;
    }
  }
  for (int k = 0; k < 32; ++k) coef[(size_t)b * 32 + k] = k < n_obs ? W[mo.reg[k] & 15u] : 0.f;
}

// PAIR: a work item takes two neighbouring amplitudes (local bit 0) with one 16-byte load and
// walks 2^4 tiles instead of 2^5 -- the same bytes in flight with half the load instructions.
template <int Q, bool PAIR, bool NT>
__global__ void __launch_bounds__(1024)
k_reg_measure_mono(const TileArgs a, const MonoObs mo, const float *__restrict__ coef) {
  extern __shared__ float4 smem4[];
  float *red = reinterpret_cast<float *>(smem4);  // [16][32]
  const int T = a.T, tid = threadIdx.x, b = blockIdx.y;
  const OpGroup g = a.groups[0];
  const int b0 = g.bits[0], b1 = g.bits[1], b2 = g.bits[2], b3 = g.bits[3];

  const uint32_t vt = PAIR ? 2u * (uint32_t)tid : (uint32_t)tid;  // index among the work-item bits
  const uint32_t lb = ins0(ins0(ins0(ins0(vt, b0), b1), b2), b3);
  uint32_t gbase = 0;
  for (int j = 0; j < T; ++j) gbase |= ((lb >> j) & 1u) << a.tile_bits[j];
  const bool thread_ok = (lb & a.zin_local) == 0;
  const uint32_t tile0 = blockIdx.x << Q;
  const float2 *pt = a.states + ((size_t)b << a.n) + tile_base(a, tile0) + gbase;
  uint32_t ostride[Q];  // element offsets of the Q low tile-index bits
#pragma unroll
  for (int j = 0; j < Q; ++j) ostride[j] = 1u << a.outer_bits[j];

  typedef float vqf __attribute__((ext_vector_type(1 << Q)));
  vqf P, P1;
#pragma unroll
  for (int it = 0; it < (1 << Q); ++it) {
    uint32_t off = 0;
#pragma unroll
    for (int j = 0; j < Q; ++j)
      if ((it >> j) & 1) off |= ostride[j];
    const bool live = thread_ok && ((tile0 + (uint32_t)it) & a.zin_outer) == 0;
    if (PAIR) {
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) {
        if (NT) {  // read once, far more than the caches hold: keep it out of them
          typedef float v4nt __attribute__((ext_vector_type(4)));
          const v4nt v = __builtin_nontemporal_load(reinterpret_cast<const v4nt *>(pt + off));
          x = make_float4(v.x, v.y, v.z, v.w);
        } else {
          x = *reinterpret_cast<const float4 *>(pt + off);
        }
      }
      P[it] = x.x * x.x + x.y * x.y;
      P1[it] = x.z * x.z + x.w * x.w;
    } else {
      float2 x = make_float2(0.f, 0.f);
      if (live) x = pt[off];
      P[it] = norm2(x);
    }
  }
#pragma unroll
  for (int h = 1; h < (1 << Q); h <<= 1) {
#pragma unroll
    for (int i = 0; i < (1 << Q); ++i) {
      if (i & h) continue;
      const float x = P[i], y = P[i | h];
      P[i] = x + y;
      P[i | h] = x - y;
      if (PAIR) {
        const float x1 = P1[i], y1 = P1[i | h];
        P1[i] = x1 + y1;
        P1[i | h] = x1 - y1;
      }
    }
  }

  const float *cf = coef + (size_t)b * 32;
  float A[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    A[k] = 0.f;
    if (k < a.n_obs) {
      const uint32_t mout = mo.out[k], mthr = mo.thr[k];
      const uint32_t idx = mout & ((1u << Q) - 1u);  // wave-uniform register index
      float val = P[idx];
      if (PAIR) val += (mthr & 1u) ? -P1[idx] : P1[idx];  // the odd neighbour: local bit 0 set
      val *= cf[k];
      const uint32_t par = (__popc(tile0 & mout) + __popc(vt & mthr)) & 1u;
      A[k] = par ? -val : val;
    }
  }
  const int lane = tid & (kWave - 1), wv = tid / kWave, nw = (blockDim.x + kWave - 1) / kWave;
  const float mine = wave_reduce_scatter<32>(A);
  if (lane < 32) red[wv * 32 + lane] = mine;
  __syncthreads();
  float *po = reinterpret_cast<float *>(a.out) +
              ((size_t)b * gridDim.x + blockIdx.x) * (QMLE_MAX_QUBITS + 1);
  if (tid <= QMLE_MAX_QUBITS) {
    float r = 0.f;
    if (tid < a.n_obs)
      for (int w = 0; w < nw; ++w) r += red[w * 32 + tid];
    po[tid] = r;
  }
}

// ---- product pass (Stage::product_ok) -----------------------------------------------------
// First columns U_g e_0 of the stage's gate groups, one work item per (group, sample).
__global__ void __launch_bounds__(64)
k_fold_columns(const LoweredOp *__restrict__ ops, const OpGroup *__restrict__ groups, int n_groups,
               const float *__restrict__ mats, uint32_t mat_floats, float2 *__restrict__ cols,
               int batch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_groups * batch) return;
  const int b = i / n_groups, gi = i - b * n_groups;
  const OpGroup g = groups[gi];
  const float *mrow = mats + (size_t)b * mat_floats;
  float2 v[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) v[c] = make_float2(c == 0 ? 1.f : 0.f, 0.f);
  for (int k = 0; k < g.n_ops; ++k) {
    const LoweredOp op = ops[g.op_begin + k];
    const Mat2 m = load_mat2(mrow + op.mat_off);
    const int cb = op.nc ? op.c0 : -1;
    if (op.flags & LF_PERMX) reg_dispatch<2>(v, m, cb, op.t0);
    else if (op.flags & LF_DIAG) reg_dispatch<1>(v, m, cb, op.t0);
    else reg_dispatch<0>(v, m, cb, op.t0);
  }
  float2 *o = cols + (size_t)i * 16;
#pragma unroll
  for (int c = 0; c < 16; ++c) o[c] = v[c];
}

__device__ __forceinline__ uint32_t pext_mask(uint32_t x, uint32_t mask) {  // gather the bits of x under mask
  uint32_t r = 0, k = 0;
  while (mask) {
    const uint32_t low = mask & (0u - mask);
    if (x & low) r |= 1u << k;
    ++k;
    mask ^= low;
  }
  return r;
}

// out[e] = in[e with the group bits cleared] * prod_g col_g[bits of e under group g]: the live
// inputs (2^(T - 4 G) per tile) are parked in LDS first, since the pass runs in place.
__global__ void __launch_bounds__(1024)
k_tile_product(const TileArgs a, const float2 *__restrict__ cols, int tiles_per_wg, uint32_t n_tiles) {
  extern __shared__ float4 smem4[];
  const int T = a.T, L = a.L, G = a.n_groups;
  float2 *tc = reinterpret_cast<float2 *>(smem4);          // [G <= 4][16]
  float2 *lin = tc + 64;                                   // [2^(T - 4G)]
  const int n_live = T - 4 * G;
  uint32_t *lut = reinterpret_cast<uint32_t *>(lin + (1u << n_live));
  const int tid = threadIdx.x, nt = blockDim.x, b = blockIdx.y;
  tile_build_lut(a, lut);
  const uint32_t lowmask = (1u << L) - 1u;
  float2 *st = a.states + ((size_t)b << a.n);

  uint32_t gm[4] = {0u, 0u, 0u, 0u}, gm_all = 0;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    if (g < G) {
      const OpGroup og = a.groups[g];
      gm[g] = (1u << og.bits[0]) | (1u << og.bits[1]) | (1u << og.bits[2]) | (1u << og.bits[3]);
      gm_all |= gm[g];
    }
  const uint32_t livemask = ((1u << T) - 1u) & ~gm_all;
  for (int i = tid; i < G * 16; i += nt) tc[i] = cols[(size_t)b * (G * 16) + i];

  // everything below but the tile base is the same for every tile: work-item indices into the
  // live-input table and the column tables, split into the part the work item fixes (local bits
  // 1 .. T-4) and the part the iteration fixes (the 3 top local bits)
  const bool bit0_live = (livemask & 1u) != 0;
  const uint32_t jt = 2u * (uint32_t)tid;
  const uint32_t lc_t = pext_mask(jt, livemask);
  uint32_t ig_t[4], odd[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    ig_t[g] = pext_mask(jt, gm[g]);
    odd[g] = (gm[g] & 1u) ? 1u : 0u;  // bit 0 is the lowest bit of its group
  }
  uint32_t top_l[3], top_g[4][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint32_t bitv = 1u << (T - 3 + k);
    top_l[k] = pext_mask(bitv, livemask);
#pragma unroll
    for (int g = 0; g < 4; ++g) top_g[g][k] = pext_mask(bitv, gm[g]);
  }
  // live-input slots this work item fills (<= 4: 2^n_live <= 4 * blockDim)
  uint32_t in_e[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t idx = (uint32_t)tid + (uint32_t)r * (uint32_t)nt;
    uint32_t e = 0, rest = idx, m = livemask;  // deposit idx under livemask
    while (rest && m) {
      const uint32_t low = m & (0u - m);
      if (rest & 1u) e |= low;
      m ^= low;
      rest >>= 1;
    }
    in_e[r] = e;
  }
  __syncthreads();  // lut, tc

  for (int tt = 0; tt < tiles_per_wg; ++tt) {
    uint32_t tile = blockIdx.x * (uint32_t)tiles_per_wg + (uint32_t)tt;
    if (tile >= n_tiles) break;
    if (a.compact) {
      uint32_t rest = tile, free_bits = a.tile_free;
      tile = 0;
      while (rest) {
        const uint32_t low = free_bits & (0u - free_bits);
        if (rest & 1u) tile |= low;
        free_bits ^= low;
        rest >>= 1;
      }
    }
    float2 *pt = st + tile_base(a, tile);
    if ((tile & a.zin_outer) != 0) {
      // (full grid only) nothing but known zeros in, nothing but zeros out -- and the input
      // may never have been written: store the zeros without reading
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t j = jt | ((uint32_t)u << (T - 3));
        *reinterpret_cast<float4 *>(pt + (lut[j >> L] | (j & lowmask))) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      continue;
    }
    if (tt) __syncthreads();  // the previous tile's reads of lin are done
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t idx = (uint32_t)tid + (uint32_t)r * (uint32_t)nt;
      if (idx < (1u << n_live)) {
        float2 x = make_float2(0.f, 0.f);
        if ((in_e[r] & a.zin_local) == 0) x = pt[lut[in_e[r] >> L] | (in_e[r] & lowmask)];
        lin[idx] = x;
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t lc = lc_t | ((u & 1) ? top_l[0] : 0u) | ((u & 2) ? top_l[1] : 0u) |
                          ((u & 4) ? top_l[2] : 0u);
      float2 f0 = make_float2(1.f, 0.f), f1 = make_float2(1.f, 0.f);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (g < G) {
          const uint32_t ig = ig_t[g] | ((u & 1) ? top_g[g][0] : 0u) | ((u & 2) ? top_g[g][1] : 0u) |
                              ((u & 4) ? top_g[g][2] : 0u);
          const float2 c0 = tc[g * 16 + ig], c1 = tc[g * 16 + (ig | odd[g])];
          f0 = cmul(c0, f0);
          f1 = cmul(c1, f1);
        }
      const float2 x0 = lin[lc], x1 = lin[bit0_live ? (lc | 1u) : lc];
      const float2 o0 = cmul(f0, x0), o1 = cmul(f1, x1);
      const uint32_t j = jt | ((uint32_t)u << (T - 3));
      *reinterpret_cast<float4 *>(pt + (lut[j >> L] | (j & lowmask))) = make_float4(o0.x, o0.y, o1.x, o1.y);
    }
  }
}

// The same product, laid out for the memory system: a work item keeps two neighbouring live
// amplitudes in registers and walks ALL 2^(4G) values of the group bits, so every store
// instruction of a workgroup covers one contiguous 4 KiB run (k_tile_product's tile geometry
// gives 128-byte runs).  In place: the only input a work item overwrites (group bits = 0) is
// the one it holds.  Needs the compact convention (known-zero outputs are not stored).
struct ProductArgs {
  float2 *states;
  const float2 *cols;      // [batch][G][16]
  uint32_t live_mask;      // bit positions an input can be non-zero on
  uint32_t gpos[4][4];     // bit positions of group g's 4 bits (gather order)
  int n, G;
};

template <bool NT>
__global__ void __launch_bounds__(256) k_product_stream(const ProductArgs a) {
  __shared__ float2 tc[4][16];
  __shared__ uint32_t goff[4][16];
  const int tid = threadIdx.x, b = blockIdx.y;
  if (tid < 64) {
    const int g = tid >> 4, c = tid & 15;
    float2 v = make_float2(1.f, 0.f);
    uint32_t off = 0;
    if (g < a.G) {
      v = a.cols[((size_t)b * a.G + g) * 16 + c];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if ((c >> i) & 1) off |= 1u << a.gpos[g][i];
    }
    tc[g][c] = v;
    goff[g][c] = off;
  }
  // element offset of this work item's pair: deposit its compact live index under live_mask
  uint32_t rest = ((uint32_t)blockIdx.x * 256u + (uint32_t)tid) * 2u, m = a.live_mask, e = 0;
  while (rest) {
    const uint32_t low = m & (0u - m);
    if (rest & 1u) e |= low;
    m ^= low;
    rest >>= 1;
  }
  float2 *pt = a.states + ((size_t)b << a.n) + e;
  const float4 x = *reinterpret_cast<const float4 *>(pt);
  const float2 x0 = make_float2(x.x, x.y), x1 = make_float2(x.z, x.w);
  __syncthreads();
  const int n3 = a.G > 3 ? 16 : 1, n2 = a.G > 2 ? 16 : 1, n1 = a.G > 1 ? 16 : 1;
  for (int i3 = 0; i3 < n3; ++i3)
    for (int i2 = 0; i2 < n2; ++i2)
      for (int i1 = 0; i1 < n1; ++i1) {
        const float2 f123 = cmul(tc[3][i3], cmul(tc[2][i2], tc[1][i1]));
        const uint32_t o123 = goff[3][i3] | goff[2][i2] | goff[1][i1];
#pragma unroll 4
        for (int i0 = 0; i0 < 16; ++i0) {
          const float2 f = cmul(f123, tc[0][i0]);
          const float2 o0 = cmul(f, x0), o1 = cmul(f, x1);
          st4<NT>(reinterpret_cast<float4 *>(pt + (o123 | goff[0][i0])), make_float4(o0.x, o0.y, o1.x, o1.y));
        }
      }
}

// ---- prefetching tile kernel ------------------------------------------------------------
// EXPERIMENT, opt-in (QMLE_PLAN_PREFETCH): in k_tile a workgroup's HBM traffic stops while it
// runs its gate groups.  k_tile_pf gives every workgroup a contiguous run of tiles and TWO
// tile buffers: while the gate groups run on one buffer the next tile streams into the other
// by LDS-DMA (global_load_lds_dwordx4, no VGPRs), so loads are in flight all the time.
// Bit-identical to k_tile (tests), but slower on MI355X: see launch_tile and DESIGN.md 9.
//
// The DMA is issued from inline asm: hipcc drains a builtin LDS-DMA with vmcnt(0) in front of
// every ds_read (it cannot tell the buffers apart), which would serialise exactly what this
// kernel overlaps.  Ordering is therefore explicit: each wave waits for its own DMAs with a
// counted vmcnt, then a barrier publishes the tile (read a staged buffer only after the
// barrier behind the wait); barriers are raw (tile_sync<true>) so that nothing drains the
// prefetch or the previous tile's stores.
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
  unsigned keep;  // lds_dst: wave-uniform LDS byte address; lane l lands at lds_dst + 16 l
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

constexpr int kPfDmaPerWave = 8;  // 2^T * 8 B / (waves * 1 KiB) for T = 12 / 13 at 2^(T-4) threads

// Issue this wave's share of the tile's loads into the LDS buffer at byte address lds_base.
// Granule (16 B) position p of the buffer holds amplitude pair g = p ^ ((p >> 4) & 15): the
// sw() layout expressed on the SOURCE address, since the DMA destination is lane-linear.
__device__ __forceinline__ void pf_issue_tile(const TileArgs &a, const float2 *st, uint64_t base,
                                              const uint32_t *lut, uint32_t lds_base) {
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t nw = blockDim.x >> 6;
  const uint32_t lowmask = (1u << a.L) - 1u;
#pragma unroll
  for (int j = 0; j < kPfDmaPerWave; ++j) {
    const uint32_t q = (uint32_t)j * nw + w;          // 1 KiB block of the buffer
    const uint32_t p = q * 64u + lane;
    const uint32_t e = (p ^ ((p >> 4) & 15u)) << 1;   // local amplitude index (even)
    glds16(st + (base | lut[e >> a.L] | (e & lowmask)), lds_base + q * 1024u);
  }
}

template <bool DENSE4>
__global__ void __launch_bounds__(512)
k_tile_pf(const TileArgs a, uint32_t n_tiles, uint32_t total, uint32_t chunk) {
  extern __shared__ float4 smem4[];
  const int T = a.T, L = a.L;
  float2 *buf0 = reinterpret_cast<float2 *>(smem4);
  float2 *buf1 = buf0 + (1u << T);
  uint32_t *lut = reinterpret_cast<uint32_t *>(buf1 + (1u << T));
  const uint32_t lut_n = (1u << (T - L)) < 4u ? 4u : (1u << (T - L));
  float *red = reinterpret_cast<float *>(lut + lut_n);
  OpSlot *slots = reinterpret_cast<OpSlot *>(red + 288);
  const size_t D = (size_t)1 << a.n;
  const uint32_t first = blockIdx.x * chunk;
  const uint32_t last = first + chunk < total ? first + chunk : total;
  if (first >= last) return;  // whole workgroup leaves together
  const uint32_t lds0 = (uint32_t)(uintptr_t)buf0;  // low 32 bits of an LDS pointer = byte address
  const uint32_t buf_bytes = 8u << T;

  tile_build_lut(a, lut);
  __syncthreads();
  int cur = 0, staged_b = -1;
  {
    const int b = (int)(first / n_tiles);
    pf_issue_tile(a, a.states + (size_t)b * D, tile_base(a, first % n_tiles), lut, lds0);
  }
  for (uint32_t f = first; f < last; ++f) {
    const int b = (int)(f / n_tiles);
    const uint32_t tile = f % n_tiles;
    const uint64_t base = tile_base(a, tile);
    if (b != staged_b) {  // (every wave passed the previous iteration's closing barrier)
      tile_stage_slots(a, slots, b);
      staged_b = b;
    }
    if (f + 1 < last) {
      const int bn = (int)((f + 1) / n_tiles);
      pf_issue_tile(a, a.states + (size_t)bn * D, tile_base(a, (f + 1) % n_tiles), lut,
                    lds0 + (uint32_t)(cur ^ 1) * buf_bytes);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // = kPfDmaPerWave: tile f has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    tile_sync<true>();  // publishes tile f (and the slots)
    float2 *s = cur ? buf1 : buf0;
    tile_compute<DENSE4, true>(a, s, slots, b);
    tile_epilogue<true>(a, s, lut, red, tile, n_tiles, b, base);
    tile_sync<true>();  // buffer `cur` is free for the DMA of tile f + 2
    cur ^= 1;
  }
}


}  // namespace

namespace qmle {

size_t tile_lds_bytes(int T, int L, int n_slots) {
  const size_t lut_n = ((size_t)1 << (T - L)) < 4 ? 4 : ((size_t)1 << (T - L));
  return ((size_t)8 << T) + 4 * lut_n + 288 * sizeof(float) + (size_t)n_slots * sizeof(OpSlot);
}

int tile_threads(int T) {  // one register-tile work item (16 amplitudes) per thread
  int t = T >= 4 ? 1 << (T - 4) : 64;
  if (t < 64) t = 64;
  if (t > 1024) t = 1024;
  return t;
}

int launch_tile(const qmle_plan *p, const Stage &st, float2 *states, const float *mats,
                const float *angles, int batch, bool init_zero, int meas, void *out,
                const uint32_t *obs_masks, int n_obs, hipStream_t stream,
                bool from_zero, float2 *cols, int *row_shift) {
  // *row_shift: TM_EXPVAL_PARTIAL rows cover 2^row_shift tiles each (multi-tile k_tile2)
  if (row_shift) *row_shift = 0;
  from_zero = from_zero && plan_sparse(p);
  TileArgs a = fill_tile_args(p, st, states, mats, angles, init_zero, meas, out, obs_masks, n_obs,
                              from_zero);
  a.slots_in_lds = tile_lds_bytes(st.T, st.L, a.n_ops) <= 160 * 1024 ? 1 : 0;
  static const bool no_nt = std::getenv("QMLE_TILE_NO_NT") != nullptr;
  // dense stages only (a stage that skips known zeros moves a fraction of the state, and what
  // it writes is read back at once): K2 dense 122.6 -> 119.8 ms per step
  // (the initialising pass only writes, and what it writes is read back by the next pass: plain
  // stores are 1.5 us per 2^24-amplitude state faster there, 22.9 vs 24.4)
  a.nt = !no_nt && st.T < p->n && !(from_zero && st.zero_in) && !init_zero &&
                 ((uint64_t)batch << (p->n + 3)) >= (1ull << 30)
             ? 1 : 0;
  a.mw_lean = meas == TM_STORE_MW && mw_lean(p->n, st) ? 1 : 0;  // (run_mw_fused asks the same question)
  // the state this pass stores is read back by the later reads only after >= 1 GiB more has been written: streaming
  // stores keep it from lingering dirty in the Infinity Cache, where its write-back would run into the first
  // later read (measured n = 28: that read 0.41 ms behind plain stores, 0.31 ms stand-alone) -- QMLE_MW_NT=0: A/B
  // The same holds for the LAST storing pass of any run whose states exceed the caches (TM_STORE, no tile stage
  // behind it): whatever reads them next -- the stand-alone Meyer-Wallach reads, a <Z> sweep, the caller -- finds
  // HBM idle instead of a write-back in progress.  QMLE_LAST_PASS_NT=0: A/B (read per launch).
  if ((meas == TM_STORE_MW || (meas == TM_STORE && !st.next_tile && !init_zero)) && st.T < p->n && !no_nt &&
      ((uint64_t)batch << (p->n + 3)) >= (1ull << 30)) {
    const char *e = std::getenv("QMLE_LAST_PASS_NT");
    if (!e || atoi(e) != 0) a.nt = 1;
  }
  const size_t lds = tile_lds_bytes(st.T, st.L, a.slots_in_lds ? a.n_ops : 0);
  if (FirstUse once{0}; once.first) {
    QMLE_LDS_BASE_CHECK(k_tile<false>);
    QMLE_LDS_BASE_CHECK(k_tile<true>);
    QMLE_LDS_BASE_CHECK((k_tile<false, true>));
    QMLE_LDS_BASE_CHECK((k_tile<true, true>));
    HIPCHK(hipFuncSetAttribute((const void *)k_tile<false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_tile<true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_tile<false, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_tile<true, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    once.done();
  }
  bool has_dense4 = false;  // 16x16 Kraus superoperators: separate instantiation, so that the
                            // common kernel keeps its register budget
  for (int g = st.grp_begin; g < st.grp_end; ++g) has_dense4 |= p->op_groups[g].kind == GK_DENSE4 || p->op_groups[g].kind == GK_REG4X;
  const unsigned tiles = 1u << (p->n - st.T);
  // Prefetching variant: tiles are loaded (not generated), the geometry is the standard one
  // (2^(T-4) threads, 8 DMAs per wave) and every workgroup gets a run of >= 4 tiles.
  const uint64_t total = (uint64_t)tiles * (uint64_t)batch;
  const int threads = tile_threads(st.T);
  const size_t lds_pf = lds + ((size_t)8 << st.T);
  static int n_cu_of[kMaxDevices] = {};
  int &n_cu = n_cu_of[current_device()];
  if (!n_cu) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, current_device()) == hipSuccess && v > 0)
      n_cu = v;
    else
      n_cu = 256;
  }
  const unsigned wg_per_cu = (unsigned)(160 * 1024 / lds_pf);
  // Opt-in (plan flag or QMLE_PREFETCH=1): measured SLOWER than k_tile on MI355X (K2, n = 24:
  // 354 vs 248 ms per 1024 states) -- two tile buffers leave room for 2 workgroups = 2 waves
  // per SIMD, and the gate groups need >= 4 to hide their own LDS / VALU latencies.
  static const bool pf_env_on = std::getenv("QMLE_PREFETCH") != nullptr;
  const bool pf_ok = (pf_env_on || (p->flags & QMLE_PLAN_PREFETCH)) && !init_zero && meas != TM_EXPVAL && a.slots_in_lds && st.L >= 1 &&
                     (st.T == 12 || st.T == 13) && threads == (1 << (st.T - 4)) &&
                     wg_per_cu >= 1 && total < (1ull << 31) &&
                     total >= 4ull * n_cu * wg_per_cu;
  if (pf_ok) {
    if (FirstUse once{1}; once.first) {
      QMLE_LDS_BASE_CHECK(k_tile_pf<false>);
      QMLE_LDS_BASE_CHECK(k_tile_pf<true>);
      HIPCHK(hipFuncSetAttribute((const void *)k_tile_pf<false>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void *)k_tile_pf<true>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      once.done();
    }
    const unsigned wgs = (unsigned)n_cu * wg_per_cu;
    const uint32_t chunk = (uint32_t)((total + wgs - 1) / wgs);
    const unsigned gx = (unsigned)((total + chunk - 1) / chunk);
    if (has_dense4)
      hipLaunchKernelGGL(k_tile_pf<true>, dim3(gx), dim3(threads), lds_pf, stream, a, tiles,
                         (uint32_t)total, chunk);
    else
      hipLaunchKernelGGL(k_tile_pf<false>, dim3(gx), dim3(threads), lds_pf, stream, a, tiles,
                         (uint32_t)total, chunk);
    HIPCHK(hipGetLastError());
    return QMLE_OK;
  }
  dim3 grid(tiles, (unsigned)batch);
  // All-live initialising pass (no known-zero bookkeeping downstream, so every tile must be
  // stored): the zeros come from a plain fill at the rate of a fill, tile 0 of every state from
  // the tile kernel behind it (15 us per 32 states) -- 22.4 -> 20.2 us per 2^24-amplitude state.
  static const bool no_fill = std::getenv("QMLE_NO_INIT_FILL") != nullptr;
  if (init_zero && !from_zero && meas == TM_STORE && st.T < p->n && !no_fill && tiles > 1 &&
      (st.fast_ok || st.T == kLdsMaxQubits) && p->n <= 28 && threads == (1 << (st.T - 4)) &&
      !(p->flags & QMLE_PLAN_PREFETCH)) {
    launch_fill_zero(states, ((uint64_t)batch << p->n) / 2u /* float4 = two amplitudes */, stream);
    a.compact = 1;  // grid = the tiles that can be non-zero = tile 0
    a.tile_free = 0u;
    grid.x = 1u;
  } else if (st.shift) {
    return QMLE_ERR_UNSUPPORTED;  // a top-first tile exists as the filled first pass of a run from |0..0> only
  }
  if (from_zero && meas == TM_STORE && st.next_tile) {
    // the zero tiles are not even launched: the next tile stage never reads them
    const uint32_t all_outer = tiles - 1u;
    const uint32_t zo = init_zero ? all_outer : a.zin_outer;
    if (zo) {
      a.compact = 1;
      a.tile_free = all_outer & ~zo;
      grid.x = 1u << __builtin_popcount(a.tile_free);
    }
  }
  static const bool no_product = std::getenv("QMLE_NO_PRODUCT") != nullptr;
  if (from_zero && cols && st.product_ok && !init_zero && meas == TM_STORE && !no_product &&
      threads == (1 << (st.T - 4))) {
    const int G = st.grp_end - st.grp_begin;
    const int items = G * batch;
    hipLaunchKernelGGL(k_fold_columns, dim3((items + 63) / 64), dim3(64), 0, stream, p->dev.d_ops,
                       p->dev.d_op_groups + st.grp_begin, G, mats, p->mat_floats, cols, batch);
    // streaming layout when the pass may leave known-zero outputs unwritten, bit 0 is live and
    // there are at least ~128 workgroups of 512 live amplitudes
    uint32_t live = ~st.zero_in & (p->n >= 32 ? ~0u : ((1u << p->n) - 1u));
    const int n_live = __builtin_popcount(live);
    static const bool no_stream = std::getenv("QMLE_NO_PRODUCT_STREAM") != nullptr;
    uint32_t gm_global = 0;
    for (int g = 0; g < G; ++g)
      for (int i = 0; i < 4; ++i)
        gm_global |= 1u << st.tile_bits[p->op_groups[st.grp_begin + g].bits[i]];
    const bool zeros_may_stay = st.next_tile || (st.zero_in & ~gm_global) == 0;
    static const uint64_t stream_min_wgs = [] {
      const char *e = std::getenv("QMLE_STREAM_MIN_WGS");
      const long v = e ? atol(e) : 0;
      return (uint64_t)(v > 0 ? v : 128);  // K2, 32 states = 256 workgroups: 49 vs 73 us (tile layout)
    }();
    if (zeros_may_stay && (live & 1u) && n_live >= 9 && !no_stream &&
        ((uint64_t)batch << (n_live - 9)) >= stream_min_wgs) {
      ProductArgs pa;
      std::memset(&pa, 0, sizeof(pa));
      pa.states = states;
      pa.cols = cols;
      pa.live_mask = live;
      pa.n = p->n;
      pa.G = G;
      for (int g = 0; g < G; ++g)
        for (int i = 0; i < 4; ++i)
          pa.gpos[g][i] = (uint32_t)st.tile_bits[p->op_groups[st.grp_begin + g].bits[i]];
      const dim3 pgrid(1u << (n_live - 9), (unsigned)batch);
      const int n_out = n_live + 4 * G;  // amplitudes written per state = 2^n_out
      // >= 1 GiB written per launch: non-temporal stores (the pass itself is no faster, the
      // measuring pass that follows is: 3.13 -> 2.99 ms per K2 step)
      if (((uint64_t)batch << (n_out + 3)) >= (1ull << 30))
        hipLaunchKernelGGL(k_product_stream<true>, pgrid, dim3(256), 0, stream, pa);
      else
        hipLaunchKernelGGL(k_product_stream<false>, pgrid, dim3(256), 0, stream, pa);
      HIPCHK(hipGetLastError());
      return QMLE_OK;
    }
    const size_t lds_p = 64 * sizeof(float2) + ((size_t)8 << (st.T - 4 * G)) +
                         ((size_t)4 << (st.T - st.L)) + 64;
    const uint32_t n_tiles = grid.x;
    int tpw = 1;  // tiles per workgroup: the index tables are built once
    while (tpw < 8 && (uint64_t)(n_tiles / (2 * tpw)) * batch >= 2048) tpw *= 2;
    grid.x = (n_tiles + tpw - 1) / tpw;
    hipLaunchKernelGGL(k_tile_product, grid, dim3(threads), lds_p, stream, a, cols, tpw, n_tiles);
    HIPCHK(hipGetLastError());
    return QMLE_OK;
  }
  // fast path: all-live stage of (controlled) 2x2 gates -- table-addressed groups, CX folded
  // into the LDS layout, SGPR matrices (k_tile2)
  static const bool no_fast = std::getenv("QMLE_NO_FAST_TILE") != nullptr;
  // (k_tile2 addresses a tile with 32-bit byte offsets inside one state: n <= 28; a whole state
  // of 10..13 qubits is one tile per sample: T == n, <Z> through the TM_EXPVAL epilogue)
  static const bool no_fast_whole = std::getenv("QMLE_NO_FAST_WHOLE") != nullptr;
  if (!no_fast && st.fast_ok && p->n <= 28 && threads == (1 << (st.T - 4)) &&
      (st.T < p->n ? meas != TM_EXPVAL : !no_fast_whole)) {
    if (FirstUse once{2}; once.first) {
#define QMLE_T2_LDS(NT, ME, MU)                                                   \
  QMLE_LDS_BASE_CHECK((k_tile2<NT, ME, MU>));                                      \
  HIPCHK(hipFuncSetAttribute((const void *)k_tile2<NT, ME, MU>,                    \
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
      QMLE_T2_LDS(false, false, false); QMLE_T2_LDS(true, false, false);
      QMLE_T2_LDS(false, true, false); QMLE_T2_LDS(true, true, false);
      QMLE_T2_LDS(false, false, true); QMLE_T2_LDS(true, false, true);
      QMLE_T2_LDS(false, true, true); QMLE_T2_LDS(true, true, true);
#undef QMLE_T2_LDS
      QMLE_LDS_BASE_CHECK((k_tile2<true, true, true, false, false, true>));
      QMLE_LDS_BASE_CHECK((k_tile2<false, true, true, false, false, true>));
      HIPCHK(hipFuncSetAttribute((const void *)k_tile2<true, true, true, false, false, true>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void *)k_tile2<false, true, true, false, false, true>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      QMLE_LDS_BASE_CHECK((k_tile2<false, false, false, true>));
      QMLE_LDS_BASE_CHECK((k_tile2<false, true, false, true>));
      QMLE_LDS_BASE_CHECK((k_tile2<false, true, false, false, true>));
      QMLE_LDS_BASE_CHECK((k_tile2<true, true, false, false, true>));
      QMLE_LDS_BASE_CHECK((k_tile2<false, true, false, true, true>));
      HIPCHK(hipFuncSetAttribute((const void *)k_tile2<false, true, false, false, true>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void *)k_tile2<true, true, false, false, true>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void *)k_tile2<false, true, false, true, true>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void *)k_tile2<false, false, false, true>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void *)k_tile2<false, true, false, true>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      once.done();
    }
    Tile2Args f;
    f.groups = p->dev.d_groups2 + st.fast_begin;
    f.ops = p->dev.d_ops2;
    f.tbl = p->dev.d_tbl2;
    f.n_groups = st.fast_end - st.fast_begin;
    f.n_ops_stage = 0;
    for (int g = st.fast_begin; g < st.fast_end; ++g) f.n_ops_stage += p->groups2[g].n_ops;
    f.gtab = st.fast_gtab;
    {  // outer bit positions (ascending) as contiguous runs
      int r = 0;
      const int n_outer = p->n - st.T;
      for (int i = 0; i < n_outer && r <= 6;) {
        int len = 1;
        while (i + len < n_outer && st.outer_bits[i + len] == st.outer_bits[i] + len) ++len;
        if (r < 6) {
          f.run_off[r] = (uint32_t)i;
          f.run_mask[r] = len >= 32 ? 0xffffffffu : ((1u << len) - 1u);
          f.run_pos[r] = (uint32_t)st.outer_bits[i];
        }
        ++r;
        i += len;
      }
      f.n_runs = r <= 6 ? r : -1;
      for (int k = r < 6 ? r : 6; k < 6; ++k) f.run_off[k] = f.run_mask[k] = f.run_pos[k] = 0;
      // local bits 0 .. T-4 (bit 0 of 2 tid is always clear, harmless) as runs
      r = 0;
      const int top = st.T - 4;
      for (int j = 0; j <= top && r <= 4;) {
        int len = 1;
        while (j + len <= top && st.tile_bits[j + len] == st.tile_bits[j] + len) ++len;
        if (r < 4) {
          f.in_off[r] = (uint32_t)j;
          f.in_mask[r] = (1u << len) - 1u;
          f.in_pos[r] = (uint32_t)st.tile_bits[j];
        }
        ++r;
        j += len;
      }
      f.n_in_runs = r <= 4 ? r : -1;
      for (int k = r < 4 ? r : 4; k < 4; ++k) f.in_off[k] = f.in_mask[k] = f.in_pos[k] = 0;
    }
    for (unsigned u = 0; u < 8; ++u)
      f.uoff8[u] = (((u & 1u) << st.tile_bits[st.T - 3]) | (((u >> 1) & 1u) << st.tile_bits[st.T - 2]) |
                    (((u >> 2) & 1u) << st.tile_bits[st.T - 1])) << 3;
    // plain all-live stages: several consecutive tiles per workgroup (next tile prefetched into
    // registers), as long as the grid still fills the chip a few times over
    // (default 4 for storing passes, 8 for the measuring pass, whose per-workgroup reduction is
    // then shared by 8 tiles: K2 pass 3 29.2 / 24.5 / 23.8 / 23.6 us per state at 1 / 2 / 4 / 8)
    static const int tpw_env = std::getenv("QMLE_T2_TPW") ? atoi(std::getenv("QMLE_T2_TPW")) : 0;
    // (Meyer-Wallach rows keep one tile per workgroup: a walk of 4-16 tiles that carries the ~40 sums in
    // registers and reduces once was built and measured at n = 28 -- 128 VGPRs + 64 B of scratch, no room
    // for the next tile's prefetch: 792 us for the pass against 678 with a row per tile, 473 without sums)
    const int tpw_max = tpw_env > 0 ? tpw_env : (meas == TM_EXPVAL_PARTIAL || meas == TM_EXPVAL_MASKS) ? 8 : 4;
    const uint64_t min_wgs = 5120;
    f.tpw = 1;
    f.tile_stride = 0;
    // (known zeros inside the tile are fine -- the walk's loads skip them; known-zero TILES are not)
    const bool multi_zin = std::getenv("QMLE_NO_MULTI_ZIN") == nullptr;  // (read per launch: the A/B test toggles it)
    // (Z-parity observables walk too -- tile_m_accumulate -- when the caller can take rows per walk; QMLE_NO_MASKS_MULTI=1: A/B)
    static const bool no_masks_multi = std::getenv("QMLE_NO_MASKS_MULTI") != nullptr;
    const bool masks_walk = meas == TM_EXPVAL_MASKS && n_obs <= kMaskMultiObs && row_shift && st.T >= 10 && !no_masks_multi;
    if (!a.init_zero && (!a.zin_local || multi_zin) && !a.zin_outer && !a.compact && st.T < p->n &&
        (meas == TM_STORE || meas == TM_PROBS || meas == TM_EXPVAL_PARTIAL || masks_walk)) {
      // (consecutive tile indices differ in the lowest run of outer bit positions only)
      int run0 = 1;
      while (run0 < p->n - st.T && st.outer_bits[run0] == st.outer_bits[0] + run0) ++run0;
      f.tile_stride = 1u << st.outer_bits[0];
      while (f.tpw * 2 <= tpw_max && f.tpw * 2 <= (1 << run0) && grid.x % 2u == 0 &&
             (uint64_t)(grid.x / 2u) * grid.y >= min_wgs) {
        f.tpw *= 2;
        grid.x /= 2u;
      }
    }
    if ((meas == TM_EXPVAL_PARTIAL || meas == TM_EXPVAL_MASKS) && f.tpw > 1) {
      if (!row_shift || f.tpw > 8) {  // the caller must know the row layout
        grid.x *= (unsigned)f.tpw;
        f.tpw = 1;
      } else {
        *row_shift = 31 - __builtin_clz((unsigned)f.tpw);
      }
    }
    static const bool dbg_launch = std::getenv("QMLE_DBG_LAUNCH") != nullptr;
    if (dbg_launch) fprintf(stderr, "[launch_tile] T=%d init_zero=%d zin_local=%x zin_outer=%x compact=%d meas=%d tpw=%d grid=(%u,%u)\n", st.T, a.init_zero, a.zin_local, a.zin_outer, a.compact, meas, f.tpw, grid.x, grid.y);
    static const int dbg = std::getenv("QMLE_DBG_T2") ? atoi(std::getenv("QMLE_DBG_T2")) : 0;
    f.dbg = dbg;
    if (dbg & 1) f.n_groups = 0;
    // T >= 10: the per-tile epilogues' scratch fits inside the tile; the whole-state <Z> epilogue
    // reduces while amplitudes are still being read and gets its own 288 floats
    // (whole_state_expval: one float per observable and wave -- 128 B at 10 qubits instead of the
    // 1152 B of round 2's epilogue: 18-19 instead of 17 single-wave workgroups per CU)
    // (QMLE_T2_LDS_PAD=<bytes>: occupancy experiment -- fewer workgroups per CU; read per launch)
    const char *pad_env = std::getenv("QMLE_T2_LDS_PAD");
    const size_t lds2 = ((size_t)8 << st.T) + (pad_env ? (size_t)atoi(pad_env) : 0) +
                        (meas == TM_EXPVAL ? (size_t)QMLE_MAX_QUBITS * (threads >= kWave ? threads / kWave : 1) * sizeof(float) : 0);
    const bool measure = !(meas == TM_STORE || meas == TM_PROBS);
#define QMLE_T2_GO(NT, ME, MU) \
  hipLaunchKernelGGL((k_tile2<NT, ME, MU>), grid, dim3(threads), lds2, stream, a, f)
    const bool multi = f.tpw > 1;
    static const bool no_ws = std::getenv("QMLE_NO_WS_KERNEL") != nullptr;  // (A/B: the generic instantiation)
    if (meas == TM_STORE_MW || meas == TM_MW_ONLY) {
      if (st.T == p->n) hipLaunchKernelGGL((k_tile2<false, true, false, true, true>), grid, dim3(threads), lds2, stream, a, f);
      else if (a.nt) hipLaunchKernelGGL((k_tile2<true, true, false, false, true>), grid, dim3(threads), lds2, stream, a, f);
      else hipLaunchKernelGGL((k_tile2<false, true, false, false, true>), grid, dim3(threads), lds2, stream, a, f);
    } else if (st.T == p->n && !multi && !a.nt && !no_ws) {  // the whole state in one tile (10..13 qubits)
      if (measure) hipLaunchKernelGGL((k_tile2<false, true, false, true>), grid, dim3(threads), lds2, stream, a, f);
      else hipLaunchKernelGGL((k_tile2<false, false, false, true>), grid, dim3(threads), lds2, stream, a, f);
    } else if (measure && multi && meas == TM_EXPVAL_MASKS) {
      if (a.nt) hipLaunchKernelGGL((k_tile2<true, true, true, false, false, true>), grid, dim3(threads), lds2, stream, a, f);
      else hipLaunchKernelGGL((k_tile2<false, true, true, false, false, true>), grid, dim3(threads), lds2, stream, a, f);
    } else if (measure) {
      if (multi) { if (a.nt) QMLE_T2_GO(true, true, true); else QMLE_T2_GO(false, true, true); }
      else { if (a.nt) QMLE_T2_GO(true, true, false); else QMLE_T2_GO(false, true, false); }
    } else {
      if (multi) { if (a.nt) QMLE_T2_GO(true, false, true); else QMLE_T2_GO(false, false, true); }
      else { if (a.nt) QMLE_T2_GO(true, false, false); else QMLE_T2_GO(false, false, false); }
    }
#undef QMLE_T2_GO
    HIPCHK(hipGetLastError());
    return QMLE_OK;
  }
  if (meas == TM_STORE_MW || meas == TM_MW_ONLY) {
    if (st.T < 10 || threads != (1 << (st.T - 4))) return QMLE_ERR_UNSUPPORTED;  // (callers check mw_fusable)
    if (has_dense4) hipLaunchKernelGGL((k_tile<true, true>), grid, dim3(threads), lds, stream, a);
    else hipLaunchKernelGGL((k_tile<false, true>), grid, dim3(threads), lds, stream, a);
  } else if (has_dense4) hipLaunchKernelGGL(k_tile<true>, grid, dim3(threads), lds, stream, a);
  else hipLaunchKernelGGL(k_tile<false>, grid, dim3(threads), lds, stream, a);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

// k_reg_measure* takes the last pass of a <Z> run when all its gates share one register-tile
// group (expval_kernel_of, qmle_plan.cpp)
int reg_measure_kind(const qmle_plan *p, size_t si, int n_obs) {
  static const bool off = std::getenv("QMLE_NO_REG_MEASURE") != nullptr;
  if (off || n_obs < 1 || n_obs > 32) return 0;
  return expval_kernel_of(p, si, plan_sparse(p));
}

int launch_reg_measure(const qmle_plan *p, const Stage &st, int kind, float2 *states,
                              const float *mats, const float *angles, int batch, void *out,
                              const uint32_t *obs_masks, int n_obs, hipStream_t stream, int *q_out,
                              float *coef) {
  TileArgs a = fill_tile_args(p, st, states, mats, angles, false, TM_EXPVAL_MASKS, out, obs_masks,
                              n_obs, plan_sparse(p));
  a.slots_in_lds = 1;
  const int n_outer = p->n - st.T;
  // ~4096 workgroups per launch when the batch allows, at most 64 tiles per workgroup
  int q = 0;
  while (q < 6 && q < n_outer && (((uint64_t)batch << n_outer) >> (q + 1)) >= 4096) ++q;
  if (kind == 3) q = 5;
  const size_t lds = (size_t)a.n_ops * sizeof(OpSlot) + (132 + 16 * 32 + 128) * sizeof(uint32_t);
  dim3 grid(1u << (n_outer - q), (unsigned)batch);
  if (kind == 3) {
    const OpGroup &g = p->op_groups[st.grp_begin];
    MonoObs mo;
    std::memset(&mo, 0, sizeof(mo));
    for (int k = 0; k < n_obs; ++k) {
      const uint32_t m = obs_masks[k];
      int tb = 0;
      for (int j = 0; j < st.T; ++j) {
        const uint32_t bitv = (m >> st.tile_bits[j]) & 1u;
        int gi = -1;
        for (int i = 0; i < 4; ++i)
          if (g.bits[i] == j) gi = i;
        if (gi >= 0) mo.reg[k] |= (uint8_t)(bitv << gi);
        else mo.thr[k] |= bitv << tb++;
      }
      for (int j = 0; j < n_outer; ++j) mo.out[k] |= ((m >> st.outer_bits[j]) & 1u) << j;
    }
    hipLaunchKernelGGL(k_mono_coef, dim3((batch + 63) / 64), dim3(64), 0, stream, p->dev.d_ops,
                       p->dev.d_op_groups + st.grp_begin, mats, p->mat_floats, mo, n_obs, coef, batch);
    static const bool no_pair = std::getenv("QMLE_NO_MONO_PAIR") != nullptr;
    if (!no_pair && g.bits[0] != 0 && st.tile_bits[0] == 0 && !(a.zin_local & 1u) && st.T >= 11 &&
        n_outer >= 4) {
      q = 4;
      grid.x = 1u << (n_outer - q);
      // live amplitudes per launch >= 1 GiB: stream them past the caches (0.427 -> 0.38 ms per
      // 256 states of K2; k_direct_1q's measurements say the opposite below the cache size)
      const int n_live = p->n - __builtin_popcount(st.zero_in);
      const bool nt = ((uint64_t)batch << (n_live + 3)) >= (1ull << 30);
      if (nt)
        hipLaunchKernelGGL((k_reg_measure_mono<4, true, true>), grid, dim3(1u << (st.T - 5)),
                           16 * 32 * sizeof(float), stream, a, mo, coef);
      else
        hipLaunchKernelGGL((k_reg_measure_mono<4, true, false>), grid, dim3(1u << (st.T - 5)),
                           16 * 32 * sizeof(float), stream, a, mo, coef);
    } else {
      hipLaunchKernelGGL((k_reg_measure_mono<5, false, false>), grid, dim3(1u << (st.T - 4)),
                         16 * 32 * sizeof(float), stream, a, mo, coef);
    }
  } else if (kind == 2)
    hipLaunchKernelGGL(k_reg_measure<true>, grid, dim3(1u << (st.T - 4)), lds, stream, a, q);
  else
    hipLaunchKernelGGL(k_reg_measure<false>, grid, dim3(1u << (st.T - 4)), lds, stream, a, q);
  HIPCHK(hipGetLastError());
  *q_out = q;
  return QMLE_OK;
}

}  // namespace qmle
