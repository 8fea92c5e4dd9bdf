// libqmle_sv: gfx950 (MI355X / CDNA4) statevector kernels + C-ABI entry points.
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

// ---------------------------------------------------------------------------
// per-sample gate matrices  (operations.py:1002-1045, 1053-1100, 1171-1243,
// 1255-1351, 1357-1487 -- matrices restated, evaluated in fp64, stored fp32)
// ---------------------------------------------------------------------------
struct cd {
  double re, im;
};
__device__ __forceinline__ cd cdmul(cd a, cd b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}

__device__ __forceinline__ cd cdadd(cd a, cd b) { return {a.re + b.re, a.im + b.im}; }
struct M2 {
  cd a, b, c, d;  // [[a, b], [c, d]]
};
// the 2x2 source gates of source_matrix(), entries in registers
template <class AT, class CT>
__device__ __forceinline__ M2 source_2x2(const BuildOp &b, const AT *__restrict__ ang, const CT *__restrict__ consts) {
  const cd z = {0.0, 0.0}, one = {1.0, 0.0};
  double th = 0.0, c = 1.0, s = 0.0;
  if (b.slot[0] >= 0) {
    th = (double)ang[b.slot[0]];
    sincos(0.5 * th, &s, &c);
  }
  const double r2 = 0.70710678118654752440;
  switch (b.opcode) {
    case QMLE_OP_X: case QMLE_OP_CX: case QMLE_OP_CCX: return {z, one, one, z};
    case QMLE_OP_Y: case QMLE_OP_CY: return {z, {0, -1}, {0, 1}, z};
    case QMLE_OP_Z: case QMLE_OP_CZ: return {one, z, z, {-1, 0}};
    case QMLE_OP_H: return {{r2, 0}, {r2, 0}, {r2, 0}, {-r2, 0}};
    case QMLE_OP_S: return {one, z, z, {0, 1}};
    case QMLE_OP_RX: case QMLE_OP_CRX: return {{c, 0}, {0, -s}, {0, -s}, {c, 0}};
    case QMLE_OP_RY: case QMLE_OP_CRY: return {{c, 0}, {-s, 0}, {s, 0}, {c, 0}};
    case QMLE_OP_RZ: case QMLE_OP_CRZ: return {{c, -s}, z, z, {c, s}};
    case QMLE_OP_CPHASE: {
      double sp, cp;
      sincos(th, &sp, &cp);
      return {one, z, z, {cp, sp}};
    }
    case QMLE_OP_ROT: {  // RZ(omega) RY(theta) RZ(phi), operations.py:1234-1243
      const double phi = (double)ang[b.slot[0]], theta = (double)ang[b.slot[1]], omega = (double)ang[b.slot[2]];
      double st, ct, sp, cp, sm, cm;
      sincos(0.5 * theta, &st, &ct);
      sincos(0.5 * (phi + omega), &sp, &cp);
      sincos(0.5 * (phi - omega), &sm, &cm);
      return {{cp * ct, -sp * ct}, {-cm * st, -sm * st}, {cm * st, -sm * st}, {cp * ct, sp * ct}};
    }
    case QMLE_OP_MAT1:
      return {{(double)consts[b.const_off + 0], (double)consts[b.const_off + 1]},
              {(double)consts[b.const_off + 2], (double)consts[b.const_off + 3]},
              {(double)consts[b.const_off + 4], (double)consts[b.const_off + 5]},
              {(double)consts[b.const_off + 6], (double)consts[b.const_off + 7]}};
    default: return {one, z, z, one};
  }
}

template <class AT, class CT>
__device__ void source_matrix(const BuildOp &b, const AT *__restrict__ ang,
                              const CT *__restrict__ consts, cd *M, int dim) {
  const int nn = dim * dim;
  for (int i = 0; i < nn; ++i) M[i] = {0.0, 0.0};
  double th = 0.0, c = 1.0, s = 0.0;
  if (b.slot[0] >= 0) {
    th = (double)ang[b.slot[0]];
    sincos(0.5 * th, &s, &c);
  }
  const double r2 = 0.70710678118654752440;
  switch (b.opcode) {
    case QMLE_OP_X: case QMLE_OP_CX: case QMLE_OP_CCX:
      M[1] = {1, 0}; M[2] = {1, 0}; break;
    case QMLE_OP_Y: case QMLE_OP_CY:
      M[1] = {0, -1}; M[2] = {0, 1}; break;
    case QMLE_OP_Z: case QMLE_OP_CZ:
      M[0] = {1, 0}; M[3] = {-1, 0}; break;
    case QMLE_OP_H:
      M[0] = {r2, 0}; M[1] = {r2, 0}; M[2] = {r2, 0}; M[3] = {-r2, 0}; break;
    case QMLE_OP_S:
      M[0] = {1, 0}; M[3] = {0, 1}; break;
    case QMLE_OP_RX: case QMLE_OP_CRX:  // c I - i s X
      M[0] = {c, 0}; M[1] = {0, -s}; M[2] = {0, -s}; M[3] = {c, 0}; break;
    case QMLE_OP_RY: case QMLE_OP_CRY:  // c I - i s Y
      M[0] = {c, 0}; M[1] = {-s, 0}; M[2] = {s, 0}; M[3] = {c, 0}; break;
    case QMLE_OP_RZ: case QMLE_OP_CRZ:  // diag(c - i s, c + i s)
      M[0] = {c, -s}; M[3] = {c, s}; break;
    case QMLE_OP_CPHASE: {              // diag(1, e^{i phi}) on the target
      double sp, cp;
      sincos(th, &sp, &cp);
      M[0] = {1, 0}; M[3] = {cp, sp}; break;
    }
    case QMLE_OP_ROT: {  // RZ(omega) RY(theta) RZ(phi), operations.py:1234-1243
      const double phi = (double)ang[b.slot[0]], theta = (double)ang[b.slot[1]],
                   omega = (double)ang[b.slot[2]];
      double st, ct, sp, cp, sm, cm;
      sincos(0.5 * theta, &st, &ct);
      sincos(0.5 * (phi + omega), &sp, &cp);
      sincos(0.5 * (phi - omega), &sm, &cm);
      M[0] = {cp * ct, -sp * ct};
      M[1] = {-cm * st, -sm * st};
      M[2] = {cm * st, -sm * st};
      M[3] = {cp * ct, sp * ct};
      break;
    }
    case QMLE_OP_SWAP: case QMLE_OP_CSWAP:
      M[0] = {1, 0}; M[6] = {1, 0}; M[9] = {1, 0}; M[15] = {1, 0}; break;
    case QMLE_OP_RXX:  // c I - i s X(x)X : anti-diagonal
      for (int i = 0; i < 4; ++i) { M[i * 4 + i] = {c, 0}; M[i * 4 + (3 - i)] = {0, -s}; }
      break;
    case QMLE_OP_RYY:  // Y(x)Y = antidiag(-1, 1, 1, -1)
      for (int i = 0; i < 4; ++i) {
        M[i * 4 + i] = {c, 0};
        const double sg = (i == 0 || i == 3) ? -1.0 : 1.0;
        M[i * 4 + (3 - i)] = {0, -s * sg};
      }
      break;
    case QMLE_OP_RZZ:  // diag(e^{-i t/2}, e^{+}, e^{+}, e^{-})
      M[0] = {c, -s}; M[5] = {c, s}; M[10] = {c, s}; M[15] = {c, -s}; break;
    case QMLE_OP_RZX:  // Z(x)X = [[X,0],[0,-X]]
      for (int i = 0; i < 4; ++i) M[i * 4 + i] = {c, 0};
      M[1] = {0, -s}; M[4] = {0, -s}; M[11] = {0, s}; M[14] = {0, s};
      break;
    case QMLE_OP_MAT1: case QMLE_OP_MAT2:
      for (int i = 0; i < nn; ++i)
        M[i] = {(double)consts[b.const_off + 2 * i], (double)consts[b.const_off + 2 * i + 1]};
      break;
    default:  // identity
      for (int i = 0; i < dim; ++i) M[i * dim + i] = {1, 0};
      break;
  }
}

// AT / CT / OT: angle table, constant blob and matrix row types -- float / float / float for the
// complex64 engine, double throughout for the complex128 one (qmle_run_batch_f64)
template <class AT, class CT, class OT>
__device__ __forceinline__ void build_matrices_body(const BuildOp *__restrict__ build,
                                                    const BuildGroup *__restrict__ groups, int n_groups,
                                                    const AT *__restrict__ angles, int n_slots,
                                                    const CT *__restrict__ consts, OT *__restrict__ mats,
                                                    uint32_t mat_floats) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (g >= n_groups) return;
  const BuildGroup grp = groups[g];
  const AT *ang = angles + (size_t)b * n_slots;
  const int dim = (int)grp.dim;
  if (dim == 2) {
    // 2x2 groups (all but the two-qubit Pauli rotations / SWAP / explicit 4x4): four named
    // entries in registers -- the generic path below indexes its arrays at run time and lives in
    // scratch, which made this kernel a third of the GPU time of the LDS-resident regime
    // (C4: 0.37 of 1.1 ms per 65536 states; profiles/r03_lds_regime_anatomy.txt)
    M2 M = source_2x2(build[grp.begin], ang, consts);
    for (uint32_t k = grp.begin + 1; k < grp.end; ++k) {
      const M2 S = source_2x2(build[k], ang, consts);  // later gate on the left: M <- S M
      const cd a = cdadd(cdmul(S.a, M.a), cdmul(S.b, M.c)), bb = cdadd(cdmul(S.a, M.b), cdmul(S.b, M.d));
      const cd c = cdadd(cdmul(S.c, M.a), cdmul(S.d, M.c)), d = cdadd(cdmul(S.c, M.b), cdmul(S.d, M.d));
      M = {a, bb, c, d};
    }
    OT *out = mats + (size_t)b * mat_floats + grp.mat_off;
    out[0] = (OT)M.a.re; out[1] = (OT)M.a.im; out[2] = (OT)M.b.re; out[3] = (OT)M.b.im;
    out[4] = (OT)M.c.re; out[5] = (OT)M.c.im; out[6] = (OT)M.d.re; out[7] = (OT)M.d.im;
    return;
  }
  cd M[16], S[16], R[16];
  source_matrix(build[grp.begin], ang, consts, M, dim);
  for (uint32_t k = grp.begin + 1; k < grp.end; ++k) {
    source_matrix(build[k], ang, consts, S, dim);
    for (int r = 0; r < dim; ++r)
      for (int c = 0; c < dim; ++c) {
        cd acc = {0, 0};
        for (int x = 0; x < dim; ++x) {
          const cd t = cdmul(S[r * dim + x], M[x * dim + c]);
          acc.re += t.re;
          acc.im += t.im;
        }
        R[r * dim + c] = acc;
      }
    for (int i = 0; i < dim * dim; ++i) M[i] = R[i];
  }
  OT *out = mats + (size_t)b * mat_floats + grp.mat_off;
  for (int i = 0; i < dim * dim; ++i) {
    out[2 * i] = (OT)M[i].re;
    out[2 * i + 1] = (OT)M[i].im;
  }
}
__global__ void k_build_matrices(const BuildOp *__restrict__ build,
                                 const BuildGroup *__restrict__ groups, int n_groups,
                                 const float *__restrict__ angles, int n_slots,
                                 const float *__restrict__ consts, float *__restrict__ mats,
                                 uint32_t mat_floats) {
  build_matrices_body<float, float, float>(build, groups, n_groups, angles, n_slots, consts, mats, mat_floats);
}
__global__ void k_build_matrices_f64(const BuildOp *__restrict__ build,
                                     const BuildGroup *__restrict__ groups, int n_groups,
                                     const double *__restrict__ angles, int n_slots,
                                     const double *__restrict__ consts, double *__restrict__ mats,
                                     uint32_t mat_floats) {
  build_matrices_body<double, double, double>(build, groups, n_groups, angles, n_slots, consts, mats, mat_floats);
}

// ---------------------------------------------------------------------------
// LDS tile kernel
// ---------------------------------------------------------------------------
enum TileMeas : int {
  TM_STORE = 0,   // write the tile back into the state buffer
  TM_PROBS = 1,   // write |psi|^2 to out (float)
  TM_EXPVAL = 2,  // whole-state only: <Z> on obs bits
  TM_EXPVAL_PARTIAL = 3,  // last pass of a tiled state: per-tile signed sums for EVERY bit
                          // -> out[b][tile][33] (k_expval_final reduces); state not stored
  TM_EXPVAL_MASKS = 4,    // same, for Z-parity observables (obs_mask): per-tile Walsh-Hadamard
                          // transform of |psi|^2 -> out[b][tile][k < n_obs]
};

// LDS layout of a tile: amplitude e lives in slot sw(e).  XOR-ing bits 1..4 with bits
// 5..8 keeps (even, odd) pairs adjacent (float4 staging) and spreads the 16-amplitude
// register gathers of low-bit groups over the banks (<= 2-way instead of 16-way).
__device__ __forceinline__ uint32_t sw(uint32_t e) { return e ^ (((e >> 5) & 15u) << 1); }

// ---- register-tile appliers: a[16] = amplitudes over 4 group bits, static indexing ----
template <int TB, int MODE>  // MODE 0 dense, 1 diagonal, 2 Pauli-X swap
__device__ __forceinline__ void reg_1q(float2 (&a)[16], const Mat2 &m) {
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if (c & (1 << TB)) continue;
    if (MODE == 0) {
      apply2(m, a[c], a[c | (1 << TB)]);
    } else if (MODE == 1) {
      a[c] = cmul(m.m00, a[c]);
      a[c | (1 << TB)] = cmul(m.m11, a[c | (1 << TB)]);
    } else {
      const float2 t = a[c];
      a[c] = a[c | (1 << TB)];
      a[c | (1 << TB)] = t;
    }
  }
}
template <int CB, int TB, int MODE>
__device__ __forceinline__ void reg_c1q(float2 (&a)[16], const Mat2 &m) {
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if ((c & (1 << TB)) || !(c & (1 << CB))) continue;
    if (MODE == 0) {
      apply2(m, a[c], a[c | (1 << TB)]);
    } else if (MODE == 1) {
      a[c] = cmul(m.m00, a[c]);
      a[c | (1 << TB)] = cmul(m.m11, a[c | (1 << TB)]);
    } else {
      const float2 t = a[c];
      a[c] = a[c | (1 << TB)];
      a[c | (1 << TB)] = t;
    }
  }
}
template <int MODE>
__device__ __forceinline__ void reg_dispatch(float2 (&a)[16], const Mat2 &m, int cb, int tb) {
  if (cb < 0) {
    switch (tb) {
      case 0: reg_1q<0, MODE>(a, m); break;
      case 1: reg_1q<1, MODE>(a, m); break;
      case 2: reg_1q<2, MODE>(a, m); break;
      default: reg_1q<3, MODE>(a, m); break;
    }
    return;
  }
  switch (cb * 4 + tb) {
    case 1: reg_c1q<0, 1, MODE>(a, m); break;
    case 2: reg_c1q<0, 2, MODE>(a, m); break;
    case 3: reg_c1q<0, 3, MODE>(a, m); break;
    case 4: reg_c1q<1, 0, MODE>(a, m); break;
    case 6: reg_c1q<1, 2, MODE>(a, m); break;
    case 7: reg_c1q<1, 3, MODE>(a, m); break;
    case 8: reg_c1q<2, 0, MODE>(a, m); break;
    case 9: reg_c1q<2, 1, MODE>(a, m); break;
    case 11: reg_c1q<2, 3, MODE>(a, m); break;
    case 12: reg_c1q<3, 0, MODE>(a, m); break;
    case 13: reg_c1q<3, 1, MODE>(a, m); break;
    default: reg_c1q<3, 2, MODE>(a, m); break;
  }
}

// Op descriptor + its per-sample 2x2 matrix, staged in LDS by the tile prologue so the
// gate loop never waits on dependent scalar loads from global memory.
struct OpSlot {
  LoweredOp op;
  float m[8];
};
static_assert(sizeof(OpSlot) == 48, "OpSlot layout");

// One GK_REG4 group: gather 16 amplitudes per work item, apply every op, scatter.
template <bool SLOTS>
__device__ __forceinline__ void lds_apply_group(float2 *__restrict__ s, int T, const OpGroup g,
                                                const LoweredOp *__restrict__ ops,
                                                const float *__restrict__ mrow,
                                                const OpSlot *__restrict__ slots, int op_base,
                                                uint32_t zmask = 0) {
  const int b0 = g.bits[0], b1 = g.bits[1], b2 = g.bits[2], b3 = g.bits[3];
  // sw() is linear over XOR and base & off == 0, so slot(base | off[c]) = sw(base) ^ sw(off[c]):
  // 16 wave-uniform constants + ONE v_xor per gathered amplitude
  uint32_t off[16];
#pragma unroll
  for (int c = 0; c < 16; ++c)
    off[c] = sw(((c & 1) ? (1u << b0) : 0u) | ((c & 2) ? (1u << b1) : 0u) |
                ((c & 4) ? (1u << b2) : 0u) | ((c & 8) ? (1u << b3) : 0u));
  const uint32_t cnt = 1u << (T - 4);
  for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
    const uint32_t lbase = ins0(ins0(ins0(ins0(i, b0), b1), b2), b3);
    if (lbase & zmask) continue;  // all 16 amplitudes are known zeros (TileArgs::zin_local)
    const uint32_t base = sw(lbase);
    float2 a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = s[base ^ off[c]];
    for (int k = 0; k < g.n_ops; ++k) {
      LoweredOp op;
      Mat2 m;
      if (SLOTS) {
        const OpSlot *sl = slots + (g.op_begin - op_base + k);
        op = sl->op;
        m = load_mat2(sl->m);
      } else {
        op = ops[g.op_begin + k];
        m = load_mat2(mrow + op.mat_off);
      }
      const int cb = op.nc ? op.c0 : -1;
      if (op.flags & LF_PERMX) reg_dispatch<2>(a, m, cb, op.t0);
      else if (op.flags & LF_DIAG) reg_dispatch<1>(a, m, cb, op.t0);
      else reg_dispatch<0>(a, m, cb, op.t0);
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) s[base ^ off[c]] = a[c];
  }
}

// GK_DENSE4: one dense 16x16 operator on 4 tile-local bits (2-qubit Kraus superoperator on
// vec(rho)).  Matrix rows/cols are already in gather order (host-permuted).
__device__ void lds_apply_dense4(float2 *__restrict__ s, int T, const OpGroup g,
                                 const float *__restrict__ mat) {
  const int b0 = g.bits[0], b1 = g.bits[1], b2 = g.bits[2], b3 = g.bits[3];
  uint32_t off[16];
#pragma unroll
  for (int c = 0; c < 16; ++c)
    off[c] = sw(((c & 1) ? (1u << b0) : 0u) | ((c & 2) ? (1u << b1) : 0u) |
                ((c & 4) ? (1u << b2) : 0u) | ((c & 8) ? (1u << b3) : 0u));
  const uint32_t cnt = 1u << (T - 4);
  for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
    const uint32_t base = sw(ins0(ins0(ins0(ins0(i, b0), b1), b2), b3));
    float2 a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = s[base ^ off[c]];
    // the 16 slots of a work item are private to it and already in registers: results can
    // be written back row by row
#pragma unroll 1
    for (int r = 0; r < 16; ++r) {  // rolled: one matrix row (32 scalars) live at a time
      const float *row = mat + 32 * r;
      float2 acc = make_float2(0.f, 0.f);
#pragma unroll
      for (int c = 0; c < 16; ++c) acc = cfma(make_float2(row[2 * c], row[2 * c + 1]), a[c], acc);
      const uint32_t o = ((r & 1) ? (1u << b0) : 0u) | ((r & 2) ? (1u << b1) : 0u) |
                         ((r & 4) ? (1u << b2) : 0u) | ((r & 8) ? (1u << b3) : 0u);
      s[base ^ sw(o)] = acc;
    }
  }
}

struct TileArgs {
  float2 *states;           // [B][2^n] (read unless init_zero; written for TM_STORE)
  const float *mats;        // [B][mat_floats]
  const float *angles;      // [B][n_slots]
  const float *consts;
  const LoweredOp *ops;     // the plan's op array (groups index into it)
  const OpGroup *groups;    // this stage's groups
  int n_groups;
  int op_begin;             // first op of this stage in `ops`
  int slots_in_lds;         // 1: prologue stages op descriptors + matrices in LDS
  void *out;                // TM_PROBS: float [B][2^n]; TM_EXPVAL: float [B][n_obs]
  uint32_t mat_floats;
  int n_ops, n, T, L, n_slots;
  int init_zero, meas, n_obs;
  // known-zero input (Stage::zero_in, runs from |0..0>): amplitudes whose local index meets
  // zin_local, and whole tiles whose index meets zin_outer, are exactly zero and never read.
  // compact: the grid holds only the tiles that can be non-zero (blockIdx.x has the bits of
  // tile_free deposited); the others are neither computed nor stored -- the next stage knows.
  uint32_t zin_local, zin_outer, tile_free;
  int compact;
  int nt;  // the launch streams >= 1 GiB of states: non-temporal tile loads / stores
  int8_t tile_bits[QMLE_MAX_QUBITS];
  int8_t outer_bits[QMLE_MAX_QUBITS];
  uint32_t obs_mask[QMLE_MAX_QUBITS];  // per observable: bit p set <=> Z on bit position p
  // TM_EXPVAL_PARTIAL, full-size tiles: where thread q finds <Z> of global bit position q among the
  // per-wave sums: 0..5 lane bit, 6..9 iteration bit, 10 total (q = 32), 16 + k wave-index bit k,
  // 32 + i outer position i (sign = tile-index bit i), 64 unused
  uint8_t qsrc[QMLE_MAX_QUBITS + 1];
};

__device__ __forceinline__ void sort3(int &a, int &b, int &c) {
  int t;
  if (a > b) { t = a; a = b; b = t; }
  if (b > c) { t = b; b = c; c = t; }
  if (a > b) { t = a; a = b; b = t; }
}

// Apply one lowered op to the 2^T amplitudes in LDS.  All threads participate.
__device__ void lds_apply(float2 *__restrict__ s, int T, const LoweredOp op,
                          const float *__restrict__ mrow, const float *__restrict__ consts,
                          const float *__restrict__ ang) {
  const int tid = threadIdx.x, nt = blockDim.x;
  if (op.kind == LK_1Q) {
    const Mat2 m = load_mat2(mrow + op.mat_off);
    const uint32_t tb = 1u << op.t0;
    if (op.nc == 0) {
      const uint32_t cnt = 1u << (T - 1);
      if (op.flags & LF_DIAG) {
        for (uint32_t i = tid; i < cnt; i += nt) {
          const uint32_t j0 = ins0(i, op.t0), j1 = j0 | tb;
          s[sw(j0)] = cmul(m.m00, s[sw(j0)]);
          s[sw(j1)] = cmul(m.m11, s[sw(j1)]);
        }
      } else {
        for (uint32_t i = tid; i < cnt; i += nt) {
          const uint32_t j0 = ins0(i, op.t0), j1 = j0 | tb;
          float2 a0 = s[sw(j0)], a1 = s[sw(j1)];
          apply2(m, a0, a1);
          s[sw(j0)] = a0;
          s[sw(j1)] = a1;
        }
      }
    } else {
      int p0 = op.t0, p1 = op.c0, p2 = op.nc == 2 ? op.c1 : 127;
      sort3(p0, p1, p2);
      const uint32_t cm = (1u << op.c0) | (op.nc == 2 ? (1u << op.c1) : 0u);
      const uint32_t cnt = 1u << (T - 1 - op.nc);
      const bool diag = op.flags & LF_DIAG;
      for (uint32_t i = tid; i < cnt; i += nt) {
        uint32_t j0 = ins0(ins0(i, p0), p1);
        if (op.nc == 2) j0 = ins0(j0, p2);
        j0 |= cm;
        const uint32_t j1 = j0 | tb;
        float2 a0 = s[sw(j0)], a1 = s[sw(j1)];
        if (diag) {
          a0 = cmul(m.m00, a0);
          a1 = cmul(m.m11, a1);
        } else {
          apply2(m, a0, a1);
        }
        s[sw(j0)] = a0;
        s[sw(j1)] = a1;
      }
    }
  } else if (op.kind == LK_2Q) {
    const float *mm = mrow + op.mat_off;
    float2 M[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) M[i] = make_float2(mm[2 * i], mm[2 * i + 1]);
    int p0 = op.t0, p1 = op.t1, p2 = op.nc ? op.c0 : 127;
    sort3(p0, p1, p2);
    const uint32_t cm = op.nc ? (1u << op.c0) : 0u;
    const uint32_t b0 = 1u << op.t0, b1 = 1u << op.t1;
    const uint32_t cnt = 1u << (T - 2 - op.nc);
    for (uint32_t i = tid; i < cnt; i += nt) {
      uint32_t j = ins0(ins0(i, p0), p1);
      if (op.nc) j = ins0(j, p2);
      j |= cm;
      const float2 a0 = s[sw(j)], a1 = s[sw(j | b1)], a2 = s[sw(j | b0)], a3 = s[sw(j | b0 | b1)];
      float2 r[4];
#pragma unroll
      for (int row = 0; row < 4; ++row) {
        float2 acc = cmul(M[row * 4 + 0], a0);
        acc = cfma(M[row * 4 + 1], a1, acc);
        acc = cfma(M[row * 4 + 2], a2, acc);
        acc = cfma(M[row * 4 + 3], a3, acc);
        r[row] = acc;
      }
      s[sw(j)] = r[0];
      s[sw(j | b1)] = r[1];
      s[sw(j | b0)] = r[2];
      s[sw(j | b0 | b1)] = r[3];
    }
  } else {  // LK_DIAG_ALL (whole-state tile only: local index == global index)
    const float x = ang[op.slot];
    const float *marks = consts + op.mat_off;
    const uint32_t cnt = 1u << T;
    for (uint32_t j = tid; j < cnt; j += nt) {
      float sn, cs;
      sincosf(marks[j] * x, &sn, &cs);
      s[sw(j)] = cmul(make_float2(cs, -sn), s[sw(j)]);
    }
  }
}

// Workgroup barrier.  RAW: bare s_barrier behind an LDS-only wait -- no fence, so neither
// outstanding global stores nor LDS-DMA prefetches in flight are drained (k_tile_pf);
// the caller orders its LDS-DMA explicitly.
template <bool RAW> __device__ __forceinline__ void tile_sync() {
  if (RAW) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  } else {
    __syncthreads();
  }
}

// Global bit positions of the tile's high local bits: lut[h] for h = local index >> L.
__device__ __forceinline__ void tile_build_lut(const TileArgs &a, uint32_t *lut) {
  for (uint32_t h = threadIdx.x; h < (1u << (a.T - a.L)); h += blockDim.x) {
    uint32_t v = 0;
    for (int i = 0; i < a.T - a.L; ++i) v |= ((h >> i) & 1u) << a.tile_bits[a.L + i];
    lut[h] = v;
  }
}

// Op descriptors + sample b's 2x2 matrices -> LDS.
__device__ __forceinline__ void tile_stage_slots(const TileArgs &a, OpSlot *slots, int b) {
  const float *mrow0 = a.mats + (size_t)b * a.mat_floats;
  for (int k = threadIdx.x; k < a.n_ops; k += blockDim.x) {
    const LoweredOp o = a.ops[a.op_begin + k];
    slots[k].op = o;
    if (o.kind == LK_1Q) {
      const float4 lo4 = *reinterpret_cast<const float4 *>(mrow0 + o.mat_off);
      const float4 hi4 = *reinterpret_cast<const float4 *>(mrow0 + o.mat_off + 4);
      *reinterpret_cast<float4 *>(slots[k].m) = lo4;
      *reinterpret_cast<float4 *>(slots[k].m + 4) = hi4;
    }
  }
}

__device__ __forceinline__ uint64_t tile_base(const TileArgs &a, uint32_t tile) {
  uint64_t base = 0;
  for (int i = 0; i < a.n - a.T; ++i) base |= (uint64_t)((tile >> i) & 1u) << a.outer_bits[i];
  return base;
}

// All gate groups of the stage on the tile in `s`; ends with a barrier.
template <bool DENSE4, bool RAW>
__device__ __forceinline__ void tile_compute(const TileArgs &a, float2 *s, const OpSlot *slots,
                                             int b) {
  const int T = a.T;
  const float *mrow = a.mats + (size_t)b * a.mat_floats;
  const float *ang = a.angles + (size_t)b * a.n_slots;
  // local bits still known-zero: work items holding only zeros rest (never set for k_tile_pf)
  uint32_t z = RAW ? 0u : a.zin_local;
  for (int gi = 0; gi < a.n_groups; ++gi) {
    const OpGroup g = a.groups[gi];
    if (g.kind == GK_REG4) {
      const uint32_t gm = (1u << g.bits[0]) | (1u << g.bits[1]) | (1u << g.bits[2]) | (1u << g.bits[3]);
      if (a.slots_in_lds) lds_apply_group<true>(s, T, g, a.ops, mrow, slots, a.op_begin, z & ~gm);
      else lds_apply_group<false>(s, T, g, a.ops, mrow, slots, a.op_begin, z & ~gm);
      z &= ~gm;
    } else if (DENSE4 && g.kind == GK_DENSE4) {
      lds_apply_dense4(s, T, g, a.consts + a.ops[g.op_begin].mat_off);
      z = 0;
    } else {
      lds_apply(s, T, a.ops[g.op_begin], mrow, a.consts, ang);
      z = 0;
    }
    tile_sync<RAW>();
  }
}

// Store / measure the finished tile.  n_tiles = tiles per state.
template <bool RAW, bool PARTIAL_ONLY = false>  // PARTIAL_ONLY: a.meas is TM_EXPVAL_PARTIAL (k_tile2's
                                                 // multi-tile instantiation keeps its register budget)
__device__ __forceinline__ void tile_epilogue(const TileArgs &a, float2 *s, const uint32_t *lut,
                                              float *red, uint32_t tile, uint32_t n_tiles, int b,
                                              uint64_t base, int qsrc_of_thread = -1) {
  const int T = a.T, L = a.L;
  const int tid = threadIdx.x, nt = blockDim.x;
  const size_t D = (size_t)1 << a.n;
  const uint32_t half = 1u << (T - 1);
  const uint32_t lowmask = (1u << L) - 1u;
  float2 *st = a.states + (size_t)b * D;
  if (!PARTIAL_ONLY && a.meas == TM_STORE) {
    if ((half % (8u * nt)) == 0) {
      for (uint32_t j0 = tid; j0 < half; j0 += 8u * nt) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<float4 *>(s)[sw((j0 + u * nt) * 2u) >> 1];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const uint32_t j = (j0 + u * nt) * 2u;
          float4 *dst = reinterpret_cast<float4 *>(st + (base | lut[j >> L] | (j & lowmask)));
          if (a.nt) st4<true>(dst, v[u]);
          else st4<false>(dst, v[u]);
        }
      }
    } else {
      for (uint32_t jc = tid; jc < half; jc += nt) {
        const uint32_t j = jc * 2u;
        *reinterpret_cast<float4 *>(st + (base | lut[j >> L] | (j & lowmask))) =
            reinterpret_cast<float4 *>(s)[sw(j) >> 1];
      }
    }
  } else if (!PARTIAL_ONLY && a.meas == TM_PROBS) {
    float *po = reinterpret_cast<float *>(a.out) + (size_t)b * D;
    for (uint32_t jc = tid; jc < half; jc += nt) {
      const uint32_t j = jc * 2u;
      const uint64_t g = base | lut[j >> L] | (j & lowmask);
      const float4 v = reinterpret_cast<float4 *>(s)[sw(j) >> 1];
      *reinterpret_cast<float2 *>(po + g) = make_float2(v.x * v.x + v.y * v.y, v.z * v.z + v.w * v.w);
    }
  } else if (PARTIAL_ONLY || a.meas == TM_EXPVAL_PARTIAL) {
    // element e = tid + it * nt: bits [0, tb) come from tid, the top bits from `it`
    float *po = reinterpret_cast<float *>(a.out) +
                ((size_t)b * n_tiles + tile) * (QMLE_MAX_QUBITS + 1);
    const uint32_t cnt = 1u << T;
    if (PARTIAL_ONLY || cnt == 16u * nt) {  // (k_tile2 always has 16 amplitudes per work item)
      const int qsrc = qsrc_of_thread >= 0 ? qsrc_of_thread : tid <= QMLE_MAX_QUBITS ? (int)a.qsrc[tid] : 64;
      // |amplitude|^2 of the 16 elements a lane owns, then a pruned Walsh-Hadamard butterfly over
      // the 4 iteration bits: the total and the four single-bit signed sums in 41 additions
      // (sw() is linear over XOR and nt a power of two: one address per lane, 16 wave-uniform
      // offsets -- not 16 adds + swizzles)
      float pr[16];
      // (opaque copy of the thread index: inside k_tile2's tile loop hipcc would otherwise hoist
      // the 16 addresses and the six lane-bit masks out of the loop and keep ~30 registers live
      // across the gates)
      uint32_t tid_e = (uint32_t)tid;
      asm volatile("" : "+v"(tid_e));
      const uint32_t e0 = (sw(tid_e) << 3) + lds_offset_of(s);
#pragma unroll
      for (int h = 0; h < 16; h += 8) {  // 8 reads in flight (hipcc would keep 3, to save registers)
        u64 amp[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) amp[it] = lds_ld64(e0 ^ (sw((uint32_t)(h + it) << (T - 4)) << 3));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < 8; ++it)
          pr[h + it] = norm2(make_float2(__uint_as_float((uint32_t)amp[it]), __uint_as_float((uint32_t)(amp[it] >> 32))));
        __builtin_amdgcn_sched_barrier(0);
      }
      float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3, tot;
      float s1[8], s2[4], s3[2];
#pragma unroll
      for (int i = 0; i < 8; ++i) { s1[i] = pr[2 * i] + pr[2 * i + 1]; h0 += pr[2 * i] - pr[2 * i + 1]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) { s2[i] = s1[2 * i] + s1[2 * i + 1]; h1 += s1[2 * i] - s1[2 * i + 1]; }
#pragma unroll
      for (int i = 0; i < 2; ++i) { s3[i] = s2[2 * i] + s2[2 * i + 1]; h2 += s2[2 * i] - s2[2 * i + 1]; }
      tot = s3[0] + s3[1];
      h3 = s3[0] - s3[1];
      // per wave: the total, the six lane-bit signed totals and h0..h3 through DPP wave sums
      // (66 v_add_f32_dpp, nothing on the LDS crossbar); wave-index bits are signed afterwards
      const int lane = (int)(tid_e & (kWave - 1)), w = tid / kWave, nw = (nt + kWave - 1) / kWave;
      float v[11];
#pragma unroll
      for (int j = 0; j < 6; ++j) v[j] = ((lane >> j) & 1) ? -tot : tot;
      v[6] = h0; v[7] = h1; v[8] = h2; v[9] = h3; v[10] = tot;
      wave_sums_dpp63(v);
      tile_sync<RAW>();  // k_tile2 keeps `red` INSIDE the tile buffer (32 KiB per workgroup = 5
                         // workgroups per CU): every amplitude must have been read by now
      if (lane == kWave - 1) {
#pragma unroll
        for (int j = 0; j < 11; ++j) red[w * 11 + j] = v[j];
      }
      tile_sync<RAW>();
      // thread q < n assembles <Z> of global bit position q itself (qsrc[q], filled on the host:
      // which of the 11 per-wave sums, or which tile-index bit for an outer position), thread 32
      // the total: no staging row, no serial walk over the position arrays
      if (tid <= QMLE_MAX_QUBITS) {
        const int src = qsrc;  // loaded before the reduction (a per-thread read of the arguments)
        float r = 0.f;
        if (src < 16) {                     // lane bit 0..5 -> sums 0..5; iteration bit -> 6..9; total -> 10
          for (int i = 0; i < nw; ++i) r += red[i * 11 + src];
        } else if (src < 32) {              // wave-index bit (src - 16)
          for (int i = 0; i < nw; ++i) r += ((i >> (src - 16)) & 1) ? -red[i * 11 + 10] : red[i * 11 + 10];
        } else if (src < 64) {              // outer position: tile-index bit (src - 32)
          for (int i = 0; i < nw; ++i) r += red[i * 11 + 10];
          if ((tile >> (src - 32)) & 1u) r = -r;
        }
        po[tid] = r;                        // src >= 64: unused position -> 0
      }
    } else {  // small tiles (forced geometries in tests): one reduction per local bit
      float acc_t = 0.f;
      for (int j = 0; j < T; ++j) {
        float acc = 0.f;
        for (uint32_t e = tid; e < cnt; e += nt) {
          const float pr = norm2(s[sw(e)]);
          acc += ((e >> j) & 1u) ? -pr : pr;
          if (j == 0) acc_t += pr;
        }
        const float r = block_sum(acc, red);
        if (tid == 0) po[a.tile_bits[j]] = r;
      }
      const float r = block_sum(acc_t, red);
      if (tid == 0) {
        po[QMLE_MAX_QUBITS] = r;
        for (int i = 0; i < a.n - T; ++i) po[a.outer_bits[i]] = ((tile >> i) & 1u) ? -r : r;
      }
    }
  } else if (a.meas == TM_EXPVAL_MASKS) {
    float *po = reinterpret_cast<float *>(a.out) +
                ((size_t)b * n_tiles + tile) * (QMLE_MAX_QUBITS + 1);
    const uint32_t cnt = 1u << T;
    if (cnt == 16u * nt && nt >= 64) {
      // element e = tid + it * nt: lane = local bits 0..5, wave = bits 6..T-5, it = top 4 bits.
      // Walsh-Hadamard transform of the tile's probabilities over the 4 iteration bits (in
      // registers) and the 6 lane bits (cross-lane butterflies): afterwards lane l, register
      // i of wave w holds sum_{lane', it} (-1)^{|lane' & l| + |it & i|} p(w, lane', it), i.e.
      // EVERY parity over those 10 bits at once; the observables pick theirs.
      float w[16];
#pragma unroll
      for (int it = 0; it < 16; ++it) w[it] = norm2(s[sw(tid + it * nt)]);
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
      const int lane = tid & (kWave - 1), wv = tid / kWave, nw = nt / kWave;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const bool hi = (lane >> j) & 1;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float o = __shfl_xor(w[i], 1 << j, kWave);
          w[i] = hi ? o - w[i] : w[i] + o;
        }
      }
      tile_sync<RAW>();  // all amplitudes have been read: the tile buffer becomes scratch
      float *C = reinterpret_cast<float *>(s);
#pragma unroll
      for (int i = 0; i < 16; ++i) C[(wv * 16 + i) * kWave + lane] = w[i];
      tile_sync<RAW>();
      if (tid < a.n_obs) {
        const uint32_t m = a.obs_mask[tid];
        uint32_t ml = 0, mw = 0, mi = 0, par = 0;
        for (int j = 0; j < T; ++j) {
          const uint32_t bitv = (m >> a.tile_bits[j]) & 1u;
          if (j < 6) ml |= bitv << j;
          else if (j < T - 4) mw |= bitv << (j - 6);
          else mi |= bitv << (j - (T - 4));
        }
        for (int i = 0; i < a.n - T; ++i) par ^= ((m >> a.outer_bits[i]) & 1u) & ((tile >> i) & 1u);
        float r = 0.f;
        for (int v = 0; v < nw; ++v) {
          const float c = C[(v * 16 + (int)mi) * kWave + (int)ml];
          r += (__popc((uint32_t)v & mw) & 1) ? -c : c;
        }
        po[tid] = par ? -r : r;
      }
    } else {  // small tiles (forced geometries in tests): one reduction per observable
      for (int k = 0; k < a.n_obs; ++k) {
        const uint32_t m = a.obs_mask[k];
        uint32_t mloc = 0, par = 0;
        for (int j = 0; j < T; ++j) mloc |= ((m >> a.tile_bits[j]) & 1u) << j;
        for (int i = 0; i < a.n - T; ++i) par ^= ((m >> a.outer_bits[i]) & 1u) & ((tile >> i) & 1u);
        float acc = 0.f;
        for (uint32_t e = tid; e < cnt; e += nt) {
          const float pr = norm2(s[sw(e)]);
          acc += (__popc(e & mloc) & 1) ? -pr : pr;
        }
        const float r = block_sum(acc, red);
        if (tid == 0) po[k] = par ? -r : r;
      }
    }
  } else {  // TM_EXPVAL, T == n
    float *eo = reinterpret_cast<float *>(a.out) + (size_t)b * a.n_obs;
    const uint32_t cnt = 1u << T;
    for (int k = 0; k < a.n_obs; ++k) {
      const uint32_t m = a.obs_mask[k];
      float acc = 0.f;
      for (uint32_t j = tid; j < cnt; j += nt) {
        const float pr = norm2(s[sw(j)]);
        acc += (__popc(j & m) & 1) ? -pr : pr;
      }
      const float tot = block_sum(acc, red);
      if (tid == 0) eo[k] = tot;
    }
  }
}

// One tile per workgroup: load -> gate groups -> store / measure.
template <bool DENSE4>
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
    if (a.meas == TM_STORE) {
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
    } else {  // TM_EXPVAL_PARTIAL / TM_EXPVAL_MASKS rows (TM_EXPVAL has a single tile)
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
  tile_epilogue<false>(a, s, lut, red, tile, gridDim.x, b, base);
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

// Plan data and per-sample matrices are written before the launch and never during it: reading
// them through the constant address space lets wave-uniform accesses compile to scalar loads
// (s_load_dwordx4/x8/x16 into SGPRs) instead of vector loads + v_readfirstlane.
#define QMLE_CONSTANT __attribute__((address_space(4)))
template <class X>
__device__ __forceinline__ const X QMLE_CONSTANT *as_constant(const X *p) {
  return (const X QMLE_CONSTANT *)(uintptr_t)p;
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
__device__ __forceinline__ void tile2_groups(uint32_t sb, uint32_t addr, const Tile2Args &f,
                                             const u64 QMLE_CONSTANT *mrow, int tid, bool use_skip) {
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
    const bool busy = !use_skip || __builtin_amdgcn_ballot_w64(!(addr & 1u)) != 0ull;
    addr = (addr & ~7u) + sb;
    A16 r;
    if (busy || relayout) {  // (a relayout stores every slot of the new layout, zeros included)
#define QMLE_LD(c) r.v##c = lds_ld64(addr ^ QMLE_OFF(c, o1, o2, o4, o8));
      QMLE_X16(QMLE_LD)
#undef QMLE_LD
    }
    // the 16 slot addresses are re-derived for the scatter (16 v_xor) instead of living in 16
    // VGPRs across the gates: the kernel stays within 96 VGPRs = 5 waves per SIMD
    asm volatile("" : "+v"(addr));
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
        if (busy) {
          if (f.dbg & 8) f_dense<1>(r, M0);
          else fast_dispatch(r, (int)(w0.y >> 24), M0);
        }
        continue;
      }
      const u64 QMLE_CONSTANT *mn = mrow + (w1.z >> 1);
      const Mat2S Mn = {mn[0], mn[1], mn[2], mn[3]};
      const v4u w2 = op[k + 2 < last ? k + 2 : last];
      if (busy) fast_dispatch(r, (int)(w0.y >> 24), M0);
      w0 = w1;
      w1 = w2;
      M0 = Mn;
    }
    if (relayout) {
      __syncthreads();  // every gather of the group is done: slots may change owners
      addr_next = (addr_next & ~7u) + sb;
      const uint32_t q1 = grp->off_out[1], q2 = grp->off_out[2], q4 = grp->off_out[4], q8 = grp->off_out[8];
#define QMLE_ST(c) lds_st64(addr_next ^ QMLE_OFF(c, q1, q2, q4, q8), r.v##c);
      QMLE_X16(QMLE_ST)
#undef QMLE_ST
      if (more) addr_next = f.tbl[grp[1].tbl + tid];
    } else if (busy) {
#define QMLE_ST(c) lds_st64(addr ^ QMLE_OFF(c, o1, o2, o4, o8), r.v##c);
      QMLE_X16(QMLE_ST)
#undef QMLE_ST
    }
    addr = addr_next;
    hdr = hdr_n;
    o1 = n1; o2 = n2; o4 = n4; o8 = n8;
    __syncthreads();
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
template <bool NT, bool MEASURE, bool MULTI>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(5))) k_tile2(const TileArgs a, const Tile2Args f) {
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
    if (!MEASURE && a.meas == TM_STORE) {
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

    tile2_groups(sbo, addr, f, mrow, tid, a.zin_local != 0);  // known zeros: Stage::zero_in

    if (MEASURE && MULTI) {  // (TM_EXPVAL_PARTIAL only: launch_tile)
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
  if (MEASURE && MULTI && !(f.dbg & 2))
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
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void reg_apply_group(float2 (&v)[16], const OpGroup &g,
                                                const OpSlot *slots, int op_base) {
  for (int k = 0; k < g.n_ops; ++k) {
    const OpSlot *sl = slots + (g.op_begin - op_base + k);
    const LoweredOp op = sl->op;
    const Mat2 m = load_mat2(sl->m);
    const int cb = op.nc ? op.c0 : -1;
    if (op.flags & LF_PERMX) reg_dispatch<2>(v, m, cb, op.t0);
    else if (op.flags & LF_DIAG) reg_dispatch<1>(v, m, cb, op.t0);
    else reg_dispatch<0>(v, m, cb, op.t0);
  }
}

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

// ---- whole-circuit adjoint in LDS (n <= 13) ----------------------------------------------
// One workgroup per sample keeps psi AND lambda in LDS: forward circuit (fused gate groups),
// lambda = (sum_k w_k Z_k) psi, then for every gate of the reversed, daggered tape the
// generator overlap and the inverse gate on both vectors.  No HBM traffic beyond the angle
// tables and the gradient row; one launch instead of ~(2 gates + 2 angles) launches.
struct AdjTermDev {
  int32_t out_slot;
  uint32_t xmask, zmask, pmask;  // bit positions
  int32_t n_y;
  float coef;
  int32_t marks_off;             // offset in the reverse plan's const blob, or -1
  int32_t pad;
};
struct AdjLdsArgs {
  TileArgs fwd;                  // the forward plan's whole-state stage
  const LoweredOp *rev_ops;      // reverse tape, global bit positions, one gate each
  const AdjTermDev *terms;
  int n_rev;
  const float *rev_mats;
  uint32_t rev_mat_floats;
  const float *rev_angles;
  int rev_n_slots;
  const float *rev_consts;
  const float *weights;          // [B][n_obs]
  uint32_t zmask[QMLE_MAX_QUBITS];
  int n_obs;
  float *grad;
  int n_grad_slots;
};

template <bool DENSE4>
__global__ void k_adjoint_lds(const AdjLdsArgs a) {
  extern __shared__ float4 smem4[];
  const int T = a.fwd.T;
  float2 *psi = reinterpret_cast<float2 *>(smem4);
  float2 *lam = psi + (1u << T);
  float *red = reinterpret_cast<float *>(lam + (1u << T));
  OpSlot *slots = reinterpret_cast<OpSlot *>(red + 288);
  const int tid = threadIdx.x, nt = blockDim.x;
  const int b = blockIdx.y;
  const uint32_t cnt = 1u << T;

  if (a.fwd.slots_in_lds) tile_stage_slots(a.fwd, slots, b);
  for (uint32_t e = tid; e < cnt; e += nt) psi[e] = make_float2(0.f, 0.f);
  __syncthreads();
  if (tid == 0) psi[sw(0)] = make_float2(1.f, 0.f);
  __syncthreads();
  tile_compute<DENSE4, false>(a.fwd, psi, slots, b);

  const float *w = a.weights + (size_t)b * a.n_obs;
  for (uint32_t i = tid; i < cnt; i += nt) {
    float d = 0.f;
    for (int o = 0; o < a.n_obs; ++o) d += (__popc(i & a.zmask[o]) & 1) ? -w[o] : w[o];
    const float2 v = psi[sw(i)];
    lam[sw(i)] = make_float2(d * v.x, d * v.y);
  }
  __syncthreads();

  const float *mrow = a.rev_mats + (size_t)b * a.rev_mat_floats;
  const float *ang = a.rev_angles + (size_t)b * a.rev_n_slots;
  for (int r = 0; r < a.n_rev; ++r) {
    const AdjTermDev t = a.terms[r];
    if (t.out_slot >= 0) {
      const float *marks = t.marks_off >= 0 ? a.rev_consts + t.marks_off : nullptr;
      float re = 0.f, im = 0.f;
      for (uint32_t i = tid; i < cnt; i += nt) {
        if ((i & t.pmask) != t.pmask) continue;
        const uint32_t j = i ^ t.xmask;
        const float2 l = lam[sw(i)], p = psi[sw(j)];
        const float sgn = marks ? marks[i] : ((__popc(j & t.zmask) & 1) ? -1.f : 1.f);
        re += sgn * (l.x * p.x + l.y * p.y);
        im += sgn * (l.x * p.y - l.y * p.x);
      }
      const float sr = block_sum(re, red);
      const float si = block_sum(im, red);
      if (tid == 0) {
        const int q = t.n_y & 3;
        const float v = q == 0 ? si : q == 1 ? sr : q == 2 ? -si : -sr;
        a.grad[(size_t)b * a.n_grad_slots + t.out_slot] = t.coef * v;
      }
    }
    const LoweredOp o = a.rev_ops[r];
    lds_apply(psi, T, o, mrow, a.rev_consts, ang);
    lds_apply(lam, T, o, mrow, a.rev_consts, ang);
    __syncthreads();
  }
}

// ---- fused adjoint tile pass (n >= 14) ----------------------------------------------------
// The backward sweep with the forward path's machinery: a pass stages the SAME tile of psi and
// of lambda in LDS, walks the register-tile groups of the reversed, daggered tape, and for every
// gate first takes the generator overlap Im <lambda| G |psi> on the 16 + 16 amplitudes a thread
// holds (G = X / Y / Z on the target, restricted to control = 1; P1 = |1><1| for CPhase), then
// applies the inverse gate to both.  One HBM round trip of the two states per ~20 gates instead
// of one per gate plus one per angle.
//   LoweredOp::slot (unused by 1-qubit ops) carries the stage-local index of the derivative
//   (-1: none), LoweredOp::pad the generator type.
enum AdjGen : int { AG_NONE = 0, AG_X = 1, AG_Y = 2, AG_Z = 3, AG_P1 = 4 };

// Im <y| G |x> on the 16 + 16 amplitudes of a register tile: G x by the gate appliers themselves
// (the generator as a 2x2 "gate"), so the only gate-shaped code in the sweep is reg_dispatch.
// With a control the applier leaves the control = 0 rows alone; those are taken out again with
// a second application whose matrix is zero ((Pbar + P G) x - Pbar x = P G x).
__device__ __forceinline__ Mat2 gen_matrix(int gtype) {
  Mat2 g;
  const float2 z = make_float2(0.f, 0.f), one = make_float2(1.f, 0.f);
  g.m00 = g.m01 = g.m10 = g.m11 = z;
  if (gtype == AG_X) { g.m01 = one; g.m10 = one; }
  else if (gtype == AG_Y) { g.m01 = make_float2(0.f, -1.f); g.m10 = make_float2(0.f, 1.f); }
  else if (gtype == AG_Z) { g.m00 = one; g.m11 = make_float2(-1.f, 0.f); }
  else { g.m11 = one; }  // AG_P1
  return g;
}
__device__ __forceinline__ float reg_im_dot(const float2 (&y)[16], const float2 (&t)[16]) {
  float im = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) im += y[c].x * t[c].y - y[c].y * t[c].x;
  return im;
}

struct AdjTileArgs {
  TileArgs t;               // stage of the reverse plan: groups, ops, matrices of sample b
  float2 *lam;              // [B][2^n]  (t.states = psi)
  const int32_t *term_idx;  // per dev_op of the stage: stage-local derivative index or -1
  const int32_t *gtype;     // per dev_op of the stage: AdjGen
  float *partial;           // [B][tiles][n_terms]
  int n_terms;
};

__global__ void __launch_bounds__(256) k_tile_adj(const AdjTileArgs A) {
  extern __shared__ float4 smem4[];
  const TileArgs &a = A.t;
  const int T = a.T, L = a.L;
  float2 *s0 = reinterpret_cast<float2 *>(smem4);
  float2 *s1 = s0 + (1u << T);
  uint32_t *lut = reinterpret_cast<uint32_t *>(s1 + (1u << T));
  const uint32_t lut_n = (1u << (T - L)) < 4u ? 4u : (1u << (T - L));
  OpSlot *slots = reinterpret_cast<OpSlot *>(lut + lut_n);
  float *ov = reinterpret_cast<float *>(slots + a.n_ops);  // [waves][n_terms]
  const int tid = threadIdx.x, nt = blockDim.x;
  const int lane = tid & (kWave - 1), w = tid / kWave, nw = nt / kWave;
  const int b = blockIdx.y;
  const uint32_t tile = blockIdx.x;
  const size_t D = (size_t)1 << a.n;
  const uint64_t base = tile_base(a, tile);
  tile_build_lut(a, lut);
  {  // op descriptors + sample b's inverse-gate matrices + derivative bookkeeping -> LDS
    const float *mrow0 = a.mats + (size_t)b * a.mat_floats;
    for (int k = tid; k < a.n_ops; k += nt) {
      // LoweredOp as four words: {kind, flags, t0, t1}, {c0, c1, nc, pad}, mat_off, slot
      uint4 o = *reinterpret_cast<const uint4 *>(a.ops + a.op_begin + k);
      const float4 lo4 = *reinterpret_cast<const float4 *>(mrow0 + o.z);
      const float4 hi4 = *reinterpret_cast<const float4 *>(mrow0 + o.z + 4);
      o.y = (o.y & 0x00ffffffu) | ((uint32_t)A.gtype[k] << 24);  // pad  <- generator type
      o.w = (uint32_t)A.term_idx[k];                              // slot <- derivative index
      *reinterpret_cast<uint4 *>(&slots[k].op) = o;
      *reinterpret_cast<float4 *>(slots[k].m) = lo4;
      *reinterpret_cast<float4 *>(slots[k].m + 4) = hi4;
    }
    for (int k = tid; k < nw * A.n_terms; k += nt) ov[k] = 0.f;
  }
  __syncthreads();
  const uint32_t half = 1u << (T - 1), lowmask = (1u << L) - 1u;
  float2 *st0 = a.states + (size_t)b * D, *st1 = A.lam + (size_t)b * D;
  for (uint32_t jc = tid; jc < half; jc += nt) {
    const uint32_t j = jc * 2u;
    const uint64_t g = base | lut[j >> L] | (j & lowmask);
    reinterpret_cast<float4 *>(s0)[sw(j) >> 1] = *reinterpret_cast<const float4 *>(st0 + g);
    reinterpret_cast<float4 *>(s1)[sw(j) >> 1] = *reinterpret_cast<const float4 *>(st1 + g);
  }
  __syncthreads();

  for (int gi = 0; gi < a.n_groups; ++gi) {
    const OpGroup g = a.groups[gi];  // GK_REG4 only (checked on the host)
    const int b0 = g.bits[0], b1 = g.bits[1], b2 = g.bits[2], b3 = g.bits[3];
    uint32_t off[16];
#pragma unroll
    for (int c = 0; c < 16; ++c)
      off[c] = sw(((c & 1) ? (1u << b0) : 0u) | ((c & 2) ? (1u << b1) : 0u) |
                  ((c & 4) ? (1u << b2) : 0u) | ((c & 8) ? (1u << b3) : 0u));
    const uint32_t cnt = 1u << (T - 4);
    for (uint32_t i = tid; i < cnt; i += nt) {
      const uint32_t bs = sw(ins0(ins0(ins0(ins0(i, b0), b1), b2), b3));
      float2 x[16], y[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        x[c] = s0[bs ^ off[c]];
        y[c] = s1[bs ^ off[c]];
      }
      for (int k = 0; k < g.n_ops; ++k) {
        const OpSlot *sl = slots + (g.op_begin - a.op_begin + k);
        const LoweredOp op = sl->op;
        const Mat2 m = load_mat2(sl->m);
        const int cb = op.nc ? op.c0 : -1;
        if (op.slot >= 0) {
          float2 t[16];
#pragma unroll
          for (int c = 0; c < 16; ++c) t[c] = x[c];
          reg_dispatch<0>(t, gen_matrix(op.pad), cb, op.t0);
          float im = reg_im_dot(y, t);
          if (cb >= 0) {
            // rows with control = 0 are untouched by both applications -> they cancel
            Mat2 zero = gen_matrix(AG_P1);
            zero.m11 = make_float2(0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 16; ++c) t[c] = x[c];
            reg_dispatch<1>(t, zero, cb, op.t0);
            im -= reg_im_dot(y, t);
          }
          im = wave_sum(im);
          if (lane == 0) ov[w * A.n_terms + op.slot] += im;
        }
        if (op.flags & LF_PERMX) { reg_dispatch<2>(x, m, cb, op.t0); reg_dispatch<2>(y, m, cb, op.t0); }
        else if (op.flags & LF_DIAG) { reg_dispatch<1>(x, m, cb, op.t0); reg_dispatch<1>(y, m, cb, op.t0); }
        else { reg_dispatch<0>(x, m, cb, op.t0); reg_dispatch<0>(y, m, cb, op.t0); }
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        s0[bs ^ off[c]] = x[c];
        s1[bs ^ off[c]] = y[c];
      }
    }
    __syncthreads();
  }
  for (uint32_t jc = tid; jc < half; jc += nt) {
    const uint32_t j = jc * 2u;
    const uint64_t g = base | lut[j >> L] | (j & lowmask);
    *reinterpret_cast<float4 *>(st0 + g) = reinterpret_cast<float4 *>(s0)[sw(j) >> 1];
    *reinterpret_cast<float4 *>(st1 + g) = reinterpret_cast<float4 *>(s1)[sw(j) >> 1];
  }
  for (int k = tid; k < A.n_terms; k += nt) {
    float v = 0.f;
    for (int i = 0; i < nw; ++i) v += ov[i * A.n_terms + k];
    A.partial[((size_t)b * gridDim.x + tile) * A.n_terms + k] = v;
  }
}

// grad[b][slot_k] = coef_k * sum_tiles partial[b][tile][k]; one block per (state, term)
__global__ void __launch_bounds__(256)
k_adj_tile_final(const float *__restrict__ partial, int n_tiles, int n_terms,
                 const int32_t *__restrict__ slot_of, const float *__restrict__ coef_of,
                 float *__restrict__ grad, int n_grad_slots) {
  __shared__ double red[16];
  const int b = blockIdx.x, k = blockIdx.y;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_tiles; i += blockDim.x)
    acc += partial[((size_t)b * n_tiles + i) * n_terms + k];
  const double tot = block_sum_d(acc, red);
  if (threadIdx.x == 0) grad[(size_t)b * n_grad_slots + slot_of[k]] = (float)(coef_of[k] * tot);
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

// ---------------------------------------------------------------------------
// direct (HBM-streaming) controlled-2x2 kernel: one float4 = 2 amplitudes per
// access, partners found by zero-bit insertion; in place.
//   MODE 0: no control, target bit >= 1      MODE 1: no control, target bit 0
//   MODE 2: control >= 1, target >= 1        MODE 3: control >= 1, target bit 0
//   MODE 4: control bit 0, target >= 1
// ---------------------------------------------------------------------------

// One work item per thread and an exact grid: a persistent grid-stride loop measured
// 15-20 % slower for this in-place two-stream pattern (tools/k1_tune.hip).
template <int MODE, bool DIAG, bool NT>
__global__ void __launch_bounds__(256)
k_direct_1q(float4 *__restrict__ states, int n, int pt, int pc,
            const float *__restrict__ mats, uint32_t mat_floats, uint32_t mat_off,
            uint64_t items) {
  const int b = blockIdx.y;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  float4 *st = states + (size_t)b * chunks;
  const uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (MODE < 5 && k >= items) return;  // modes 5 / 6 / 7: exact grids, whole waves
  const Mat2 m = load_mat2(mats + (size_t)b * mat_floats + mat_off);
  if constexpr (MODE == 0) {
    if constexpr (DIAG) {  // items = all chunks
      float4 v = ld4<NT>(st + k);
      const float2 f = ((k >> (pt - 1)) & 1ull) ? m.m11 : m.m00;
      const float2 x = cmul(f, make_float2(v.x, v.y)), y = cmul(f, make_float2(v.z, v.w));
      st4<NT>(st + k, make_float4(x.x, x.y, y.x, y.y));
    } else {
      const uint64_t c0 = ins0_64(k, pt - 1), c1 = c0 | (1ull << (pt - 1));
      float4 v0 = ld4<NT>(st + c0), v1 = ld4<NT>(st + c1);
      float2 a0 = make_float2(v0.x, v0.y), a1 = make_float2(v1.x, v1.y);
      float2 b0 = make_float2(v0.z, v0.w), b1 = make_float2(v1.z, v1.w);
      apply2(m, a0, a1);
      apply2(m, b0, b1);
      st4<NT>(st + c0, make_float4(a0.x, a0.y, b0.x, b0.y));
      st4<NT>(st + c1, make_float4(a1.x, a1.y, b1.x, b1.y));
    }
  } else if constexpr (MODE == 1) {
    float4 v = ld4<NT>(st + k);
    float2 a0 = make_float2(v.x, v.y), a1 = make_float2(v.z, v.w);
    if constexpr (DIAG) {
      a0 = cmul(m.m00, a0);
      a1 = cmul(m.m11, a1);
    } else {
      apply2(m, a0, a1);
    }
    st4<NT>(st + k, make_float4(a0.x, a0.y, a1.x, a1.y));
  } else if constexpr (MODE == 2) {
    if constexpr (DIAG) {  // items = chunks with control bit set
      const uint64_t c = ins0_64(k, pc - 1) | (1ull << (pc - 1));
      float4 v = ld4<NT>(st + c);
      const float2 f = ((c >> (pt - 1)) & 1ull) ? m.m11 : m.m00;
      const float2 x = cmul(f, make_float2(v.x, v.y)), y = cmul(f, make_float2(v.z, v.w));
      st4<NT>(st + c, make_float4(x.x, x.y, y.x, y.y));
    } else {
      const int lo = pt < pc ? pt - 1 : pc - 1, hi = pt < pc ? pc - 1 : pt - 1;
      const uint64_t c0 = ins0_64(ins0_64(k, lo), hi) | (1ull << (pc - 1));
      const uint64_t c1 = c0 | (1ull << (pt - 1));
      float4 v0 = ld4<NT>(st + c0), v1 = ld4<NT>(st + c1);
      float2 a0 = make_float2(v0.x, v0.y), a1 = make_float2(v1.x, v1.y);
      float2 b0 = make_float2(v0.z, v0.w), b1 = make_float2(v1.z, v1.w);
      apply2(m, a0, a1);
      apply2(m, b0, b1);
      st4<NT>(st + c0, make_float4(a0.x, a0.y, b0.x, b0.y));
      st4<NT>(st + c1, make_float4(a1.x, a1.y, b1.x, b1.y));
    }
  } else if constexpr (MODE == 3) {
    const uint64_t c = ins0_64(k, pc - 1) | (1ull << (pc - 1));
    float4 v = ld4<NT>(st + c);
    float2 a0 = make_float2(v.x, v.y), a1 = make_float2(v.z, v.w);
    if constexpr (DIAG) {
      a0 = cmul(m.m00, a0);
      a1 = cmul(m.m11, a1);
    } else {
      apply2(m, a0, a1);
    }
    st4<NT>(st + c, make_float4(a0.x, a0.y, a1.x, a1.y));
  } else if constexpr (MODE == 5) {
    // uncontrolled dense gate on bit 1..6: the partner chunk sits in lane ^ 2^(pt-1) of the same
    // wave.  Every lane loads and stores contiguous float4s (coalesced like the diagonal gate)
    // and fetches the partner's through the cross-lane path; it computes its own half of the
    // pair only.  items = chunks / 2, two rows per lane.  (tools/k1_tune.hip: 0.76 -> 0.70 ms)
    const uint64_t c0 = (uint64_t)blockIdx.x * 512u + threadIdx.x, c1 = c0 + 256u;
    const bool up = (threadIdx.x >> (pt - 1)) & 1u;
    const float2 ms = up ? m.m11 : m.m00, mo = up ? m.m10 : m.m01;
    float4 v[2] = {ld4<NT>(st + c0), ld4<NT>(st + c1)};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float4 o;
      o.x = __shfl_xor(v[u].x, 1 << (pt - 1), kWave); o.y = __shfl_xor(v[u].y, 1 << (pt - 1), kWave);
      o.z = __shfl_xor(v[u].z, 1 << (pt - 1), kWave); o.w = __shfl_xor(v[u].w, 1 << (pt - 1), kWave);
      const float2 x = cfma(mo, make_float2(o.x, o.y), cmul(ms, make_float2(v[u].x, v[u].y)));
      const float2 y = cfma(mo, make_float2(o.z, o.w), cmul(ms, make_float2(v[u].z, v[u].w)));
      v[u] = make_float4(x.x, x.y, y.x, y.y);
    }
    st4<NT>(st + c0, v[0]);
    st4<NT>(st + c1, v[1]);
  } else if constexpr (MODE == 7) {
    // controlled gate with the control on bits 0..3 and the target on bits 1..6: both live inside
    // the 1 KiB a wave covers with one float4 per lane, and every 128-byte line holds both
    // control values, so all 16 D bytes move whatever the kernel does.  Stream them like the
    // diagonal gate -- one contiguous float4 per lane in, one out --, fetch the partner through
    // the cross-lane path and rewrite only the amplitudes whose control bit is set.  (A single-
    // gate LDS tile pass did this at 0.757 ms for n = 28; 8 D accounting: 0.35 -> 0.39.)
    const uint64_t c = (uint64_t)blockIdx.x * 256u + threadIdx.x;  // items = all chunks, exact grid
    const bool up = (threadIdx.x >> (pt - 1)) & 1u;
    const float2 ms = up ? m.m11 : m.m00, mo = up ? m.m10 : m.m01;
    const float4 v = ld4<NT>(st + c);
    float4 o;
    o.x = __shfl_xor(v.x, 1 << (pt - 1), kWave); o.y = __shfl_xor(v.y, 1 << (pt - 1), kWave);
    o.z = __shfl_xor(v.z, 1 << (pt - 1), kWave); o.w = __shfl_xor(v.w, 1 << (pt - 1), kWave);
    const bool lane_ctl = pc == 0 ? true : ((threadIdx.x >> (pc - 1)) & 1u) != 0;
    float2 x = make_float2(v.x, v.y), y = make_float2(v.z, v.w);
    if (lane_ctl) {
      if (pc != 0) x = cfma(mo, make_float2(o.x, o.y), cmul(ms, x));  // control bit 0: only the odd amplitude
      y = cfma(mo, make_float2(o.z, o.w), cmul(ms, y));
    }
    st4<NT>(st + c, make_float4(x.x, x.y, y.x, y.y));
  } else if constexpr (MODE == 6) {
    // uncontrolled dense gate on a high bit (>= 21): a wave takes 4 ADJACENT rows of each of the
    // two streams (4 KiB contiguous per stream), all loads of one stream first: the DRAM banks
    // see fewer alternations between the two rows 2^pt amplitudes apart.  items = pairs / 4.
    // (tools/k1_tune.hip: 0.75-0.79 -> 0.70 ms for bits 21..27)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t row0 = ((uint64_t)blockIdx.x * 4u + wave) * 4u;
    float4 v0[4], v1[4];
    uint64_t c0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { c0[u] = ins0_64((row0 + u) * 64u + lane, pt - 1); v0[u] = ld4<NT>(st + c0[u]); }
#pragma unroll
    for (int u = 0; u < 4; ++u) v1[u] = ld4<NT>(st + (c0[u] | (1ull << (pt - 1))));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float2 a0 = make_float2(v0[u].x, v0[u].y), a1 = make_float2(v1[u].x, v1[u].y);
      float2 b0 = make_float2(v0[u].z, v0[u].w), b1 = make_float2(v1[u].z, v1[u].w);
      apply2(m, a0, a1);
      apply2(m, b0, b1);
      v0[u] = make_float4(a0.x, a0.y, b0.x, b0.y);
      v1[u] = make_float4(a1.x, a1.y, b1.x, b1.y);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) st4<NT>(st + c0[u], v0[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) st4<NT>(st + (c0[u] | (1ull << (pt - 1))), v1[u]);
  } else {  // MODE 4: control is the in-chunk bit -> only the odd amplitude
    if constexpr (DIAG) {  // items = all chunks
      float4 v = ld4<NT>(st + k);
      const float2 f = ((k >> (pt - 1)) & 1ull) ? m.m11 : m.m00;
      const float2 y = cmul(f, make_float2(v.z, v.w));
      st4<NT>(st + k, make_float4(v.x, v.y, y.x, y.y));
    } else {
      const uint64_t c0 = ins0_64(k, pt - 1), c1 = c0 | (1ull << (pt - 1));
      float4 v0 = ld4<NT>(st + c0), v1 = ld4<NT>(st + c1);
      float2 b0 = make_float2(v0.z, v0.w), b1 = make_float2(v1.z, v1.w);
      apply2(m, b0, b1);
      st4<NT>(st + c0, make_float4(v0.x, v0.y, b0.x, b0.y));
      st4<NT>(st + c1, make_float4(v1.x, v1.y, b1.x, b1.y));
    }
  }
}

__global__ void __launch_bounds__(256)
k_diag_all(float4 *__restrict__ states, int n, const float *__restrict__ marks,
           const float *__restrict__ angles, int n_slots, int slot) {
  const int b = blockIdx.y;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  float4 *st = states + (size_t)b * chunks;
  const float x = angles[(size_t)b * n_slots + slot];
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < chunks; k += stride) {
    float4 v = st[k];
    float s0, c0, s1, c1;
    sincosf(marks[2 * k] * x, &s0, &c0);
    sincosf(marks[2 * k + 1] * x, &s1, &c1);
    const float2 a = cmul(make_float2(c0, -s0), make_float2(v.x, v.y));
    const float2 c = cmul(make_float2(c1, -s1), make_float2(v.z, v.w));
    st[k] = make_float4(a.x, a.y, c.x, c.y);
  }
}

__global__ void __launch_bounds__(256)
k_init_zero(float4 *__restrict__ states, int n) {
  const int b = blockIdx.y;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  float4 *st = states + (size_t)b * chunks;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < chunks; k += stride)
    st[k] = make_float4(k == 0 ? 1.f : 0.f, 0.f, 0.f, 0.f);
}

// Zero fill of `count` float4 (the all-live initialising pass: every tile but tile 0 of a state is
// zeros).  ONE plain store per thread and no loop: 6.79 TB/s on 4 GiB; four stores per thread
// 6.29, sixteen 5.71, a grid-stride loop 5.35, hipMemsetAsync 6.59, non-temporal stores a little
// below each (tools/fill_bench.hip) -- and one workgroup per 32 KiB tile inside k_tile2 6.0.
__global__ void __launch_bounds__(256) k_fill_zero(float4 *__restrict__ p, uint64_t count) {
  const uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (k < count) p[k] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---------------------------------------------------------------------------
// all-qubit <Z> in one read.  Element index bits: bit0 = position inside the
// float4 chunk, bits 1..8 = thread id, bits 9..10 = unroll slot u, bits >= 11 =
// segment id.  Every bit gets its own signed accumulator (static indexing).
// ---------------------------------------------------------------------------
constexpr int kEzThreads = 256;
constexpr int kEzUnroll = 4;
constexpr int kEzSegBits = 11;  // 1 + 8 + 2
constexpr int kEzMaxHigh = QMLE_MAX_QUBITS - kEzSegBits;

__global__ void __launch_bounds__(kEzThreads)
k_expval_partial(const float4 *__restrict__ states, int n, float *__restrict__ partial) {
  __shared__ float red[16];
  const int b = blockIdx.y, tid = threadIdx.x;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  const float4 *st = states + (size_t)b * chunks;
  const uint64_t seg_chunks = (uint64_t)kEzThreads * kEzUnroll;
  const uint64_t n_seg = (chunks + seg_chunks - 1) / seg_chunks;
  float acc_tot = 0.f, acc_b0 = 0.f, acc_u0 = 0.f, acc_u1 = 0.f;
  float acc_hi[kEzMaxHigh];
#pragma unroll
  for (int k = 0; k < kEzMaxHigh; ++k) acc_hi[k] = 0.f;
  for (uint64_t seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
    float4 v[kEzUnroll];
#pragma unroll
    for (int u = 0; u < kEzUnroll; ++u) {
      const uint64_t c = seg * seg_chunks + (uint64_t)u * kEzThreads + tid;
      v[u] = c < chunks ? st[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float pe = 0.f, po = 0.f, pu[kEzUnroll];
#pragma unroll
    for (int u = 0; u < kEzUnroll; ++u) {
      const float e = v[u].x * v[u].x + v[u].y * v[u].y;
      const float o = v[u].z * v[u].z + v[u].w * v[u].w;
      pe += e;
      po += o;
      pu[u] = e + o;
    }
    const float tot = pe + po;
    acc_tot += tot;
    acc_b0 += pe - po;
    acc_u0 += (pu[0] - pu[1]) + (pu[2] - pu[3]);
    acc_u1 += (pu[0] + pu[1]) - (pu[2] + pu[3]);
#pragma unroll
    for (int k = 0; k < kEzMaxHigh; ++k) acc_hi[k] += ((seg >> k) & 1ull) ? -tot : tot;
  }
  // partial[b][block][bit]; bit n is the plain total (norm check)
  float *out = partial + ((size_t)b * gridDim.x + blockIdx.x) * (QMLE_MAX_QUBITS + 1);
  float r;
  r = block_sum(acc_b0, red);
  if (tid == 0) out[0] = r;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    r = block_sum(((tid >> i) & 1) ? -acc_tot : acc_tot, red);
    if (tid == 0) out[1 + i] = r;
  }
  r = block_sum(acc_u0, red);
  if (tid == 0) out[9] = r;
  r = block_sum(acc_u1, red);
  if (tid == 0) out[10] = r;
#pragma unroll
  for (int k = 0; k < kEzMaxHigh; ++k) {
    r = block_sum(acc_hi[k], red);
    if (tid == 0) out[kEzSegBits + k] = r;
  }
  r = block_sum(acc_tot, red);
  if (tid == 0) out[QMLE_MAX_QUBITS] = r;
}

struct ObsBits {
  int8_t bits[QMLE_MAX_QUBITS] = {};
  // row i of observable k counts with sign (-1)^popcount(row_mask[k] & i): observables that are Z
  // on ONE position of the last tile times Z's on outer positions (bits of the tile index)
  uint32_t row_mask[QMLE_MAX_QUBITS] = {};
};

// one block per (state, observable): fp64 sum of that bit's column over all partial rows
__global__ void __launch_bounds__(256)
k_expval_final(const float *__restrict__ partial, int n_blocks, int n_obs, ObsBits obs,
               float *__restrict__ out) {
  __shared__ double red[16];
  const int b = blockIdx.x, k = blockIdx.y;
  const float *pp = partial + (size_t)b * n_blocks * (QMLE_MAX_QUBITS + 1) + obs.bits[k];
  const uint32_t rm = obs.row_mask[k];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_blocks; i += blockDim.x) {
    const double v = (double)pp[(size_t)i * (QMLE_MAX_QUBITS + 1)];
    acc += (__popc(rm & (uint32_t)i) & 1) ? -v : v;
  }
  const double tot = block_sum_d(acc, red);
  if (threadIdx.x == 0) out[(size_t)b * n_obs + k] = (float)tot;
}

// ---------------------------------------------------------------------------
// simple streaming kernels
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_probs(const float4 *__restrict__ states, float2 *__restrict__ out, uint64_t total_chunks) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total_chunks; k += stride) {
    const float4 v = states[k];
    out[k] = make_float2(v.x * v.x + v.y * v.y, v.z * v.z + v.w * v.w);
  }
}

__global__ void __launch_bounds__(256)
k_density(const float2 *__restrict__ states, float2 *__restrict__ out, int n) {
  const int b = blockIdx.y;
  const uint64_t D = (uint64_t)1 << n;
  const float2 *st = states + (size_t)b * D;
  float2 *o = out + (size_t)b * D * D;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < D * D; k += stride) {
    const uint64_t i = k >> n, j = k & (D - 1);
    const float2 a = st[i], c = st[j];
    o[k] = make_float2(a.x * c.x + a.y * c.y, a.y * c.x - a.x * c.y);  // a * conj(c)
  }
}

struct KeepBits {
  int8_t bits[QMLE_MAX_QUBITS];  // bit positions of kept wires, LSB of output first
  int n_keep;
};

// Few kept wires (<= 12): every workgroup bins its share of |a|^2 in LDS first and adds one value
// per bin to the output -- 2^n / grid terms per global atomic instead of one.  (One global float
// atomic per amplitude left partial probabilities 2e-6 .. 2e-5 apart between two runs on the
// same state at n = 18, and it is the slowest way to add.)
__global__ void __launch_bounds__(256)
k_marginal_lds(const float2 *__restrict__ states, float *__restrict__ out, int n, KeepBits kb) {
  extern __shared__ float4 smem4[];
  float *bins = reinterpret_cast<float *>(smem4);
  const int b = blockIdx.y;
  const uint32_t n_bins = 1u << kb.n_keep;
  for (uint32_t k = threadIdx.x; k < n_bins; k += blockDim.x) bins[k] = 0.f;
  __syncthreads();
  const uint64_t D = (uint64_t)1 << n;
  const float2 *st = states + (size_t)b * D;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < D; i += stride) {
    uint32_t idx = 0;
    for (int k = 0; k < kb.n_keep; ++k) idx |= (uint32_t)((i >> kb.bits[k]) & 1ull) << k;
    atomicAdd(bins + idx, norm2(st[i]));
  }
  __syncthreads();
  float *o = out + ((size_t)b << kb.n_keep);
  for (uint32_t k = threadIdx.x; k < n_bins; k += blockDim.x) {
    const float v = bins[k];
    if (v != 0.f) atomicAdd(o + k, v);
  }
}

__global__ void __launch_bounds__(256)
k_marginal(const float2 *__restrict__ states, float *__restrict__ out, int n, KeepBits kb) {
  const int b = blockIdx.y;
  const uint64_t D = (uint64_t)1 << n;
  const float2 *st = states + (size_t)b * D;
  float *o = out + ((size_t)b << kb.n_keep);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < D; i += stride) {
    uint32_t idx = 0;
    for (int k = 0; k < kb.n_keep; ++k) idx |= (uint32_t)((i >> kb.bits[k]) & 1ull) << k;
    atomicAdd(o + idx, norm2(st[i]));
  }
}

// <a|b> partial sums: partial[pair][block] = (re, im)
__global__ void __launch_bounds__(256)
k_overlap_partial(const float4 *__restrict__ states, int n, int n_pairs,
                  float2 *__restrict__ partial) {
  __shared__ float red[16];
  const int pr = blockIdx.y;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  const float4 *a = states + (size_t)pr * chunks;
  const float4 *c = states + ((size_t)pr + n_pairs) * chunks;
  float re = 0.f, im = 0.f;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < chunks; k += stride) {
    const float4 x = a[k], y = c[k];
    // conj(x) * y
    re += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    im += x.x * y.y - x.y * y.x + x.z * y.w - x.w * y.z;
  }
  const float r = block_sum(re, red);
  const float i = block_sum(im, red);
  if (threadIdx.x == 0) partial[(size_t)pr * gridDim.x + blockIdx.x] = make_float2(r, i);
}

__global__ void __launch_bounds__(256)
k_overlap_final(const float2 *__restrict__ partial, int n_blocks, int n_pairs,
                float *__restrict__ out) {
  __shared__ double red[16];
  const int pr = blockIdx.x;
  double re = 0.0, im = 0.0;
  for (int i = threadIdx.x; i < n_blocks; i += blockDim.x) {
    const float2 v = partial[(size_t)pr * n_blocks + i];
    re += v.x;
    im += v.y;
  }
  re = block_sum_d(re, red);
  im = block_sum_d(im, red);
  if (threadIdx.x == 0) out[pr] = (float)(re * re + im * im);
}

// Meyer-Wallach cross terms c_j = sum_{bit_j = 0} psi_i conj(psi_{i + 2^j}) plus the
// populations a_j, d_j; one launch per bit (v1: n reads of the state).
// ---- parameter sampler on the device ---------------------------------------------------------
// numpy's Philox4x64-10 stream (csrc/qmle_rng.cpp restates it on the host): block b of the stream
// is philox(counter = b + 1, key) -- any block on its own, one work item per block of four values.
// The arithmetic after the generator is numpy's, rounding for rounding: u = (x >> 11) * 2^-53
// (exact), low + range * u as a rounded product and a rounded sum (no fused multiply-add), cast to
// float32.  Expressibility(12 q, 1024 pairs) spent 0.4 of its 0.9 ms drawing parameters on the host.
__global__ void __launch_bounds__(256)
k_philox_uniform(uint64_t k0, uint64_t k1, uint64_t n, double low, double range, float *__restrict__ out) {
  const uint64_t blocks = (n + 3) / 4;
  for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < blocks; b += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t c0 = b + 1, c1 = 0, c2 = 0, c3 = 0, a0 = k0, a1 = k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const uint64_t M0 = 0xD2E7470EE14C6C93ull, M1 = 0xCA5A826395121157ull;
      const uint64_t hi0 = __umul64hi(M0, c0), lo0 = M0 * c0, hi1 = __umul64hi(M1, c2), lo1 = M1 * c2;
      const uint64_t n0 = hi1 ^ c1 ^ a0, n2 = hi0 ^ c3 ^ a1;
      c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
      a0 += 0x9E3779B97F4A7C15ull;
      a1 += 0xBB67AE8584CAA73Bull;
    }
    const uint64_t v[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t i = 4 * b + (uint64_t)j;
      if (i < n) {
        const double u = __dmul_rn((double)(v[j] >> 11), 1.0 / 9007199254740992.0);
        out[i] = (float)__dadd_rn(low, __dmul_rn(range, u));
      }
    }
  }
}

// partial[b][bit][block] = (re c, im c, a, d)
__global__ void __launch_bounds__(256)
k_cross_partial(const float4 *__restrict__ states, int n, int p, float4 *__restrict__ partial,
                int n_blocks) {
  __shared__ float red[16];
  const int b = blockIdx.y;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  const float4 *st = states + (size_t)b * chunks;
  float cr = 0.f, ci = 0.f, pa = 0.f, pd = 0.f;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  if (p == 0) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < chunks; k += stride) {
      const float4 v = st[k];
      cr += v.x * v.z + v.y * v.w;
      ci += v.y * v.z - v.x * v.w;
      pa += v.x * v.x + v.y * v.y;
      pd += v.z * v.z + v.w * v.w;
    }
  } else {
    const uint64_t items = chunks >> 1;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < items; k += stride) {
      const uint64_t c0 = ins0_64(k, p - 1), c1 = c0 | (1ull << (p - 1));
      const float4 x = st[c0], y = st[c1];
      cr += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      ci += x.y * y.x - x.x * y.y + x.w * y.z - x.z * y.w;
      pa += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
      pd += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
    }
  }
  const float r0 = block_sum(cr, red), r1 = block_sum(ci, red), r2 = block_sum(pa, red),
              r3 = block_sum(pd, red);
  if (threadIdx.x == 0)
    partial[((size_t)b * n + p) * n_blocks + blockIdx.x] = make_float4(r0, r1, r2, r3);
}

__global__ void __launch_bounds__(256)
k_mw_final(const float4 *__restrict__ partial, int n, int n_blocks, int batch,
           float *__restrict__ out, float *__restrict__ purities) {
  __shared__ double red[16];
  const int b = blockIdx.x;
  double sum = 0.0;
  for (int p = 0; p < n; ++p) {
    double cr = 0, ci = 0, a = 0, d = 0;
    for (int i = threadIdx.x; i < n_blocks; i += blockDim.x) {
      const float4 v = partial[((size_t)b * n + p) * n_blocks + i];
      cr += v.x; ci += v.y; a += v.z; d += v.w;
    }
    cr = block_sum_d(cr, red);
    ci = block_sum_d(ci, red);
    a = block_sum_d(a, red);
    d = block_sum_d(d, red);
    if (threadIdx.x == 0) {
      const double pur = a * a + d * d + 2.0 * (cr * cr + ci * ci);
      if (purities) purities[(size_t)b * n + (n - 1 - p)] = (float)pur;  // index by wire
      sum += pur;
    }
  }
  if (threadIdx.x == 0) out[b] = (float)(2.0 * (1.0 - sum / n));
}

__global__ void __launch_bounds__(256)
k_histogram(const float *__restrict__ values, int64_t count, int n_bins, float lo, float hi,
            int *__restrict__ counts) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float scale = (float)n_bins / (hi - lo);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    const float v = values[i];
    if (!(v >= lo) || !(v <= hi)) continue;  // numpy drops out-of-range and NaN
    int bin = (int)((v - lo) * scale);
    if (bin >= n_bins) bin = n_bins - 1;     // right edge inclusive
    // guard against rounding across an edge: edges are lo + k*(hi-lo)/n_bins
    const float w = (hi - lo) / (float)n_bins;
    if (bin > 0 && v < lo + bin * w) --bin;
    else if (bin < n_bins - 1 && v >= lo + (bin + 1) * w) ++bin;
    atomicAdd(counts + bin, 1);
  }
}


// ---- shot sampling (simulation.py:320-377) ------------------------------------------
// Philox4x32-10 counter RNG: counter = (shot pair, 0, row lo, row hi), key = seed.
struct Philox4 { uint32_t x[4]; };
__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                 uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{{c0, c1, c2, c3}};
}
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {  // uniform in (0, 1)
  return ((double)((((uint64_t)hi << 32) | lo) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

// Inclusive fp64 prefix sum of one row of probabilities per block: cdf[b][i] = sum_{j<=i} p[b][j]
__global__ void __launch_bounds__(256)
k_cdf(const float *__restrict__ probs, uint64_t D, double *__restrict__ cdf) {
  __shared__ double wsum[4];
  __shared__ double carry_s;
  const float *p = probs + (size_t)blockIdx.x * D;
  double *c = cdf + (size_t)blockIdx.x * D;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0.0;
  __syncthreads();
  for (uint64_t base = 0; base < D; base += 1024) {
    const uint64_t i0 = base + (uint64_t)threadIdx.x * 4;
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (i0 + k < D) ? (double)p[i0 + k] : 0.0;
    v[1] += v[0];
    v[2] += v[1];
    v[3] += v[2];
    double incl = v[3];  // inclusive scan of the per-thread totals across the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    double before = carry_s + (incl - v[3]);
    for (int j = 0; j < w; ++j) before += wsum[j];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i0 + k < D) c[i0 + k] = before + v[k];
    __syncthreads();
    if (threadIdx.x == 255) carry_s = before + v[3];
    __syncthreads();
  }
}

// First index with cdf[idx] >= r (numpy.searchsorted side="left").
__device__ __forceinline__ uint32_t cdf_search(const double *__restrict__ c, uint64_t D,
                                               double r) {
  uint64_t lo = 0, hi = D - 1;  // answer in [lo, hi]; cdf[D-1] = total >= r
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (c[mid] >= r) hi = mid; else lo = mid + 1;
  }
  return (uint32_t)lo;
}

constexpr int kShotsPerThread = 16;                       // 8 Philox blocks
constexpr int kShotsPerBlock = 256 * kShotsPerThread;     // 4096
constexpr int kLdsHistMax = 4096;                         // bins kept in LDS (16 KiB)

// counts[b][idx] += 1 for `shots` draws idx ~ probs[b]; grid (ceil(shots/4096), rows)
template <bool LDS_HIST>
__global__ void __launch_bounds__(256)
k_sample(const double *__restrict__ cdf, uint64_t D, int shots, uint64_t seed,
         uint64_t row_offset, int *__restrict__ counts) {
  __shared__ int hist[LDS_HIST ? kLdsHistMax : 1];
  const double *c = cdf + (size_t)blockIdx.y * D;
  int *out = counts + (size_t)blockIdx.y * D;
  if (LDS_HIST) {
    for (uint32_t i = threadIdx.x; i < D; i += 256) hist[i] = 0;
    __syncthreads();
  }
  const double total = c[D - 1];
  const uint64_t row = row_offset + blockIdx.y;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const int64_t first = (int64_t)blockIdx.x * kShotsPerBlock;
#pragma unroll 1
  for (int j = 0; j < kShotsPerThread / 2; ++j) {
    // shot pair index: consecutive threads take consecutive pairs
    const int64_t pair = first / 2 + (int64_t)j * 256 + threadIdx.x;
    if (2 * pair >= shots) break;
    const Philox4 rnd = philox4x32_10((uint32_t)pair, (uint32_t)((uint64_t)pair >> 32),
                                      (uint32_t)row, (uint32_t)(row >> 32), k0, k1);
    const uint32_t a = cdf_search(c, D, total * u53(rnd.x[0], rnd.x[1]));
    if (LDS_HIST) atomicAdd(&hist[a], 1); else atomicAdd(out + a, 1);
    if (2 * pair + 1 < shots) {
      const uint32_t b = cdf_search(c, D, total * u53(rnd.x[2], rnd.x[3]));
      if (LDS_HIST) atomicAdd(&hist[b], 1); else atomicAdd(out + b, 1);
    }
  }
  if (LDS_HIST) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < D; i += 256)
      if (hist[i]) atomicAdd(out + i, hist[i]);
  }
}

__global__ void __launch_bounds__(256)
k_counts_to_probs(const int *__restrict__ counts, uint64_t total, float inv_shots,
                  float *__restrict__ out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
    out[i] = (float)counts[i] * inv_shots;
}

// sum_i p[b][i] * d_k[sub_k(i)]: the diagonal of observable k lifted to the register
// (simulation.py:363-372).  diag_off < 0: Z-parity over the wires (no table).
struct DiagObs {
  int8_t bits[QMLE_MAX_QUBITS];  // bit positions of the observable's wires, MSB first
  int n_wires;
  int diag_off;
};
__global__ void __launch_bounds__(256)
k_probs_diag_expval(const float *__restrict__ probs, uint64_t D, const DiagObs *__restrict__ obs,
                    const float *__restrict__ diag, int n_obs, float *__restrict__ out) {
  __shared__ double red[16];
  const DiagObs ob = obs[blockIdx.y];
  const float *p = probs + (size_t)blockIdx.x * D;
  double acc = 0.0;
  for (uint64_t i = threadIdx.x; i < D; i += 256) {
    uint32_t sub = 0;
    for (int k = 0; k < ob.n_wires; ++k) sub = (sub << 1) | (uint32_t)((i >> ob.bits[k]) & 1);
    const float d = ob.diag_off < 0 ? ((__popc(sub) & 1) ? -1.f : 1.f) : diag[ob.diag_off + sub];
    acc += (double)(p[i] * d);
  }
  const double t = block_sum_d(acc, red);
  if (threadIdx.x == 0) out[(size_t)blockIdx.x * n_obs + blockIdx.y] = (float)t;
}


// ---- adjoint differentiation -------------------------------------------------------------
struct ZSumArgs {
  uint32_t mask[QMLE_MAX_QUBITS];  // bit-position parity masks
  int n_obs;
};
// lambda[b][i] = (sum_k w[b][k] (-1)^{|i & mask_k|}) psi[b][i]
__global__ void __launch_bounds__(256)
k_zsum_apply(const float4 *__restrict__ psi, float4 *__restrict__ lam, int n,
             const float *__restrict__ weights, ZSumArgs z) {
  const int b = blockIdx.y;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  const float *w = weights + (size_t)b * z.n_obs;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < chunks; k += stride) {
    const uint32_t i0 = (uint32_t)(k << 1);
    float d0 = 0.f, d1 = 0.f;
    for (int o = 0; o < z.n_obs; ++o) {
      const float wk = w[o];
      d0 += (__popc(i0 & z.mask[o]) & 1) ? -wk : wk;
      d1 += (__popc((i0 | 1u) & z.mask[o]) & 1) ? -wk : wk;
    }
    const float4 v = psi[(size_t)b * chunks + k];
    lam[(size_t)b * chunks + k] = make_float4(d0 * v.x, d0 * v.y, d1 * v.z, d1 * v.w);
  }
}

struct AdjTerm {
  uint32_t xmask, zmask, pmask;  // bit positions: flipped / sign / projected onto 1
  const float *marks;            // != nullptr: G = diag(marks)
};
// partial[b][block] = sum_i conj(lambda_i) (X^x Z^z Pi_p psi)_i   (phase i^n_y applied later)
__global__ void __launch_bounds__(256)
k_adj_overlap(const float4 *__restrict__ psi_all, const float4 *__restrict__ lam_all, int n,
              AdjTerm t, float2 *__restrict__ partial) {
  __shared__ float red[16];
  const int b = blockIdx.y;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  const float4 *psi = psi_all + (size_t)b * chunks;
  const float4 *lam = lam_all + (size_t)b * chunks;
  const uint64_t xk = t.xmask >> 1;
  const bool swap01 = t.xmask & 1u;
  float re = 0.f, im = 0.f;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < chunks; k += stride) {
    const uint32_t i0 = (uint32_t)(k << 1);
    if ((i0 & t.pmask & ~1u) != (t.pmask & ~1u)) continue;
    const float4 l = lam[k];
    float4 p = psi[k ^ xk];
    if (swap01) p = make_float4(p.z, p.w, p.x, p.y);
    // element e of this chunk: row i = i0 | e, source j = i ^ xmask
    float s0, s1;
    if (t.marks) {
      s0 = t.marks[i0];
      s1 = t.marks[i0 | 1u];
    } else {
      const uint32_t j0 = i0 ^ t.xmask, j1 = (i0 | 1u) ^ t.xmask;
      s0 = (__popc(j0 & t.zmask) & 1) ? -1.f : 1.f;
      s1 = (__popc(j1 & t.zmask) & 1) ? -1.f : 1.f;
    }
    if ((t.pmask & 1u)) s0 = 0.f;  // projector wants bit 0 = 1: even rows drop out
    re += s0 * (l.x * p.x + l.y * p.y) + s1 * (l.z * p.z + l.w * p.w);
    im += s0 * (l.x * p.y - l.y * p.x) + s1 * (l.z * p.w - l.w * p.z);
  }
  const float r = block_sum(re, red);
  const float i = block_sum(im, red);
  if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = make_float2(r, i);
}
// grad[b][slot] = coef * Im(i^n_y * sum)
__global__ void __launch_bounds__(256)
k_adj_final(const float2 *__restrict__ partial, int n_blocks, int n_y, float coef,
            float *__restrict__ grad, int n_grad_slots, int slot) {
  __shared__ double red[16];
  const int b = blockIdx.x;
  double re = 0.0, im = 0.0;
  for (int k = threadIdx.x; k < n_blocks; k += blockDim.x) {
    const float2 v = partial[(size_t)b * n_blocks + k];
    re += v.x;
    im += v.y;
  }
  const double r = block_sum_d(re, red);
  const double i = block_sum_d(im, red);
  if (threadIdx.x == 0) {
    const int q = n_y & 3;
    const double v = q == 0 ? i : q == 1 ? r : q == 2 ? -i : -r;
    grad[(size_t)b * n_grad_slots + slot] = (float)(coef * v);
  }
}


// <a_i|b_i> for separate arrays a, b: partial[i][block] = (re, im)
__global__ void __launch_bounds__(256)
k_overlap2_partial(const float4 *__restrict__ a_all, const float4 *__restrict__ b_all, int n,
                   float2 *__restrict__ partial) {
  __shared__ float red[16];
  const int pr = blockIdx.y;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  const float4 *a = a_all + (size_t)pr * chunks;
  const float4 *c = b_all + (size_t)pr * chunks;
  float re = 0.f, im = 0.f;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < chunks; k += stride) {
    const float4 x = a[k], y = c[k];
    re += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    im += x.x * y.y - x.y * y.x + x.z * y.w - x.w * y.z;
  }
  const float r = block_sum(re, red);
  const float i = block_sum(im, red);
  if (threadIdx.x == 0) partial[(size_t)pr * gridDim.x + blockIdx.x] = make_float2(r, i);
}

__global__ void __launch_bounds__(256)
k_overlap2_final(const float2 *__restrict__ partial, int n_blocks, int count,
                 float2 *__restrict__ out) {
  __shared__ double red[16];
  const int pr = blockIdx.x;
  double re = 0.0, im = 0.0;
  for (int i = threadIdx.x; i < n_blocks; i += blockDim.x) {
    const float2 v = partial[(size_t)pr * n_blocks + i];
    re += v.x;
    im += v.y;
  }
  re = block_sum_d(re, red);
  im = block_sum_d(im, red);
  if (threadIdx.x == 0) out[pr] = make_float2((float)re, (float)im);
}

// Z-parity expectation: sum_i (-1)^{popcount(i & mask)} |psi_i|^2, up to 8 masks per launch.
struct ParityMasks {
  uint32_t m[8];
  int count;
};

__global__ void __launch_bounds__(256)
k_parity_partial(const float4 *__restrict__ states, int n, ParityMasks pm,
                 float *__restrict__ partial) {
  __shared__ float red[16];
  const int b = blockIdx.y;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  const float4 *st = states + (size_t)b * chunks;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += stride) {
    const float4 v = st[c];
    const float pe = v.x * v.x + v.y * v.y, po = v.z * v.z + v.w * v.w;
    const uint32_t ie = (uint32_t)(c << 1), io = ie | 1u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc[k] += (__popc(ie & pm.m[k]) & 1) ? -pe : pe;
      acc[k] += (__popc(io & pm.m[k]) & 1) ? -po : po;
    }
  }
  float *out = partial + ((size_t)b * gridDim.x + blockIdx.x) * 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float r = block_sum(acc[k], red);
    if (threadIdx.x == 0) out[k] = r;
  }
}

__global__ void __launch_bounds__(256)
k_parity_final(const float *__restrict__ partial, int n_blocks, int count, int n_obs_total,
               int obs_off, float *__restrict__ out) {
  __shared__ double red[16];
  const int b = blockIdx.x;
  const float *pp = partial + (size_t)b * n_blocks * 8;
  for (int k = 0; k < count; ++k) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_blocks; i += blockDim.x) acc += (double)pp[(size_t)i * 8 + k];
    const double tot = block_sum_d(acc, red);
    if (threadIdx.x == 0) out[(size_t)b * n_obs_total + obs_off + k] = (float)tot;
  }
}


// ---------------------------------------------------------------------------
// Meyer-Wallach (entanglement.py:69-103 with Tr rho_j^2 = a^2 + d^2 + 2 |c|^2): THREE reads of
// the state at n = 28 -- ceil((n - 12) / 8) + 1 in general -- instead of n.
//
// A read streams the state through 2^12-amplitude LDS tiles: the 4 lowest bits (128-byte rows)
// + two runs of 4 bit positions, [lo, lo+4) and [lo2, lo2+4), and reports the cross terms
// c_j = sum_{bit_j = 0} psi_i conj(psi_{i + 2^j}) of its bits.  The FIRST read's tile is 32 KiB
// of contiguous memory (bits 0..11); it also reports the signed populations of those 12 bits and
// the per-row totals from which the populations of EVERY other bit follow (sign of a row = one
// bit of its index), so the later reads carry cross terms only.
//
// Round-3 structure (tools/mw_tune.hip has the stand-alone bench it was tuned with):
//  * a lane's own 8 float4 span local bits {0, 9, 10, 11}: their cross terms (and, first read,
//    the 4-bit population butterfly) come straight out of the load registers, before staging;
//  * then ONE staging round trip and cross-only register gathers: local bits 1..4 and 5..8
//    (first read: 2 x 16 ds_read_b64) resp. 4..7 and, for bit 8 alone, 8 ds_read_b128 over
//    {0, 8, 9, 10} (later reads) -- 192 resp. 128 packed fmas per 16 amplitudes, nothing else;
//  * populations of the thread-mapped local bits 1..8 are the per-thread totals signed by the
//    thread index, applied once in the reduction at the end of a workgroup's walk;
//  * no register prefetch: 91 / 108 VGPRs = 5 / 4 workgroups per CU keep more bytes in flight
//    than a software pipeline at 3 (the round-2 kernel: 128 - 144 VGPRs, 4.1 - 4.85 TB/s);
//  * which positions share a tile matters: at n = 28 the pairs {12-15, 24-27} and {16-19, 20-23}
//    stream at 6.9 TB/s, {12-15, 20-23} at 5.7 and {20-27} at 4.5 (same bytes, same code;
//    profiles/r03_mw_tune.txt) -- mw_plan() pairs the 4-bit chunks outermost with innermost.
// Measured at n = 28 (MI355X): first read 0.34 ms, later reads 0.31 ms each; round 2: 0.52 +
// 2 x 0.44.
// ---------------------------------------------------------------------------
constexpr int kMwT = 12, kMwThreads = 256;
constexpr int kMwRowFirst = 48, kMwRowLater = 16;

struct MwReadArgs {
  const float2 *states;
  float *rows;  // [batch][rows_per_state][kMwRowFirst | kMwRowLater]
  int n, lo, lo2, q;  // tile = bits 0..3 + lo..lo+3 + lo2..lo2+3; 2^q tiles per workgroup
};

// x conj(y) accumulated into (re, im): two packed fmas
__device__ __forceinline__ void mw_cross(v2f &s, v2f x, v2f y) {
  s = __builtin_elementwise_fma(x, y.xx, s);
  s = __builtin_elementwise_fma((v2f){x.y, -x.x}, y.yy, s);
}
// cross terms of the 4 bits a 16-amplitude register gather spans
template <int B0>
__device__ __forceinline__ void mw_cross16(v2f (&cr)[12], const v2f (&r)[16]) {
  static_for<4>([&](auto t) {
    static_for<8>([&](auto pq) {
      constexpr int lowm = (1 << t) - 1;
      constexpr int c = (((int)pq & ~lowm) << 1) | ((int)pq & lowm);
      mw_cross(cr[B0 + (int)t], r[c], r[c | (1 << t)]);
    });
  });
}

template <bool FIRST, bool NT>
__device__ __forceinline__ void mw_read_body(const MwReadArgs &a, float4 *smem4) {
  const uint32_t sbo = lds_offset_of(smem4);  // 0: no static LDS (launch side checks lds_base_is_zero)
  const uint32_t tid = threadIdx.x;
  const uint32_t jl = 2u * tid;  // local bits 1..8 from tid, 9..11 from u, bit 0 inside the float4
  const int lo = a.lo, lo2 = a.lo2;
  const uint64_t goff = ((uint64_t)(jl & 15u) | ((uint64_t)((jl >> 4) & 15u) << lo) | ((uint64_t)(jl >> 8) << lo2)) << 3;
  const uint64_t ustep = (uint64_t)1 << (lo2 + 1 + 3);  // local bit 9 = second run's bit 1
  const char *st = reinterpret_cast<const char *>(a.states + ((size_t)blockIdx.y << a.n)) + goff;
  const uint32_t slb = (sw(jl) << 3) + sbo;  // staging address of u = 0; u adds u << 12
  const uint32_t tile0 = blockIdx.x << a.q, n_it = 1u << a.q;
  const uint32_t r0 = lo - 4, r1 = lo2 - lo - 4;  // outer runs [4, lo), [lo+4, lo2), [lo2+4, n)

  // cross terms per reported bit: first read local bit b at cr[b]; later reads new bit k at cr[k]
  // (k < 4: lo + k, else lo2 + k - 4)
  v2f cr[12];
  static_for<12>([&](auto k) { cr[k] = (v2f){0.f, 0.f}; });
  float zin[4] = {0.f, 0.f, 0.f, 0.f}, tot = 0.f, zw[4] = {0.f, 0.f, 0.f, 0.f};

  for (uint32_t it = 0; it < n_it; ++it) {
    const uint32_t t = tile0 + it;
    const uint64_t base = ((uint64_t)(t & ((1u << r0) - 1u)) << 4 | (uint64_t)((t >> r0) & ((1u << r1) - 1u)) << (lo + 4) |
                           (uint64_t)(t >> (r0 + r1)) << (lo2 + 4)) << 3;
    float4 v4[8];
    static_for<8>([&](auto u) { v4[u] = ld4<NT>(reinterpret_cast<const float4 *>(st + base + (uint64_t)u * ustep)); });
    v2f lo_[8], hi_[8];  // the two amplitudes of each float4
    static_for<8>([&](auto u) { lo_[u] = (v2f){v4[u].x, v4[u].y}; hi_[u] = (v2f){v4[u].z, v4[u].w}; });
    // ---- bits held by the lane's own 8 float4: local 0 (halves of a float4) and 9, 10, 11 (u) ----
    if (FIRST) {
      static_for<8>([&](auto u) { mw_cross(cr[0], lo_[u], hi_[u]); });
      float pr[16];
      static_for<8>([&](auto u) {
        const v2f q0 = lo_[u] * lo_[u], q1 = hi_[u] * hi_[u];
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
      static_for<4>([&](auto j) { zw[j] += __uint_as_float(__float_as_uint(tt) ^ (((it >> j) & 1u) << 31)); });
    }
    static_for<3>([&](auto k) {
      constexpr int B = FIRST ? 9 + (int)k : 5 + (int)k;
      static_for<4>([&](auto pq) {
        constexpr int lowm = (1 << k) - 1;
        constexpr int u0 = (((int)pq & ~lowm) << 1) | ((int)pq & lowm);
        mw_cross(cr[B], lo_[u0], lo_[u0 | (1 << k)]);
        mw_cross(cr[B], hi_[u0], hi_[u0 | (1 << k)]);
      });
    });
    if (it) __syncthreads();  // the previous tile's gathers are done
    static_for<8>([&](auto u) { lds_st128(slb + ((uint32_t)u << 12), v4[u]); });
    __syncthreads();
    uint32_t tg = tid;
    asm volatile("" : "+v"(tg));  // keeps the gather addresses out of loop-carried registers
    {  // gather A: 16 amplitudes over local bits gA .. gA+3 (first read 1..4, later reads 4..7)
      constexpr int gA = FIRST ? 1 : 4;
      const uint32_t bs = (sw(ins0(ins0(ins0(ins0(tg, gA), gA + 1), gA + 2), gA + 3)) << 3) + sbo;
      v2f r[16];
      static_for<16>([&](auto c) {
        const u64 x = lds_ld64(bs ^ (sw((uint32_t)c << gA) << 3));
        r[c] = (v2f){__uint_as_float((uint32_t)x), __uint_as_float((uint32_t)(x >> 32))};
      });
      mw_cross16<FIRST ? gA : 0>(cr, r);
    }
    if (FIRST) {  // gather B: local bits 5..8
      constexpr int gB = 5;
      const uint32_t bs = (sw(ins0(ins0(ins0(ins0(tg, gB), gB + 1), gB + 2), gB + 3)) << 3) + sbo;
      v2f r[16];
      static_for<16>([&](auto c) {
        const u64 x = lds_ld64(bs ^ (sw((uint32_t)c << gB) << 3));
        r[c] = (v2f){__uint_as_float((uint32_t)x), __uint_as_float((uint32_t)(x >> 32))};
      });
      mw_cross16<gB>(cr, r);
    } else {  // local bit 8 alone: 8 float4 over local bits {0, 8, 9, 10}; thread index -> 1..7, 11
      const uint32_t e0 = ((tg & 127u) << 1) | ((tg >> 7) << 11);
      const uint32_t bs = (sw(e0) << 3) + sbo;
      float4 r[8];
      static_for<8>([&](auto c) {
        constexpr uint32_t e = (((uint32_t)c & 1u) << 8) | (((uint32_t)c >> 1) << 9);
        r[c] = lds_ld128(bs ^ (sw(e) << 3));
      });
      static_for<4>([&](auto pq) {
        mw_cross(cr[4], (v2f){r[2 * pq].x, r[2 * pq].y}, (v2f){r[2 * pq + 1].x, r[2 * pq + 1].y});
        mw_cross(cr[4], (v2f){r[2 * pq].z, r[2 * pq].w}, (v2f){r[2 * pq + 1].z, r[2 * pq + 1].w});
      });
    }
  }
  // ---- one reduction and one row per workgroup ----
  // first read: [0..23] cross terms of local bit b at 2b, 2b+1; [24..35] signed populations of
  // local bits 0..11; [36] total; [37..40] total signed by bit j of the tile's index in the walk.
  // later reads: [0..15] cross terms of new bit k at 2k, 2k+1.
  constexpr int NB = FIRST ? 12 : 8, NV = FIRST ? 41 : 16;
  float red_v[NV];
  static_for<NB>([&](auto k) { red_v[2 * k] = cr[k].x; red_v[2 * k + 1] = cr[k].y; });
  if (FIRST) {
    red_v[24] = zin[0];
    static_for<8>([&](auto k) { red_v[25 + k] = ((tid >> k) & 1u) ? -tot : tot; });
    red_v[33] = zin[1]; red_v[34] = zin[2]; red_v[35] = zin[3];
    red_v[36] = tot;
    static_for<4>([&](auto j) { red_v[37 + j] = zw[j]; });
  }
  wave_sums_dpp63(red_v);
  __syncthreads();  // every gather has been read: the tile becomes scratch
  float *red = reinterpret_cast<float *>(smem4);
  const uint32_t lane = tid & (kWave - 1), w = tid / kWave;
  if (lane == kWave - 1) static_for<NV>([&](auto k) { red[w * NV + k] = red_v[k]; });
  __syncthreads();
  if (tid < NV) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMwThreads / kWave; ++i) s += red[i * NV + tid];
    a.rows[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (FIRST ? kMwRowFirst : kMwRowLater) + tid] = s;
  }
}
// Two entry points, because the occupancy that suits them differs: the later reads are bound by
// HBM alone and want every workgroup the LDS admits (5 per CU); the first read carries twice the
// arithmetic and runs FASTER at 4 workgroups per CU (0.33 vs 0.42 ms at n = 28, same
// instructions -- five workgroups of it fight over the vector pipe and the LDS between barriers).
template <bool NT>
__global__ void __launch_bounds__(kMwThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) k_mw_read_first(const MwReadArgs a) {
  extern __shared__ float4 smem4[];
  mw_read_body<true, NT>(a, smem4);
}
template <bool NT>
__global__ void __launch_bounds__(kMwThreads) k_mw_read_later(const MwReadArgs a) {
  extern __shared__ float4 smem4[];
  mw_read_body<false, NT>(a, smem4);
}

// Where the sums of bit position p come from: which later read (0 = the first read) and column.
struct MwPlan {
  int n_later;
  int lo[8], lo2[8], q[8];   // later reads
  int q_first;
  int src_read[QMLE_MAX_QUBITS], src_col[QMLE_MAX_QUBITS];
  uint32_t rows_first, rows_later[8];
};

// purity of one wire from the per-workgroup rows: one block per (state, bit position)
struct MwPurityArgs {
  const float *first;       // [batch][rows_first][kMwRowFirst]
  const float *later[8];    // [batch][rows_later[r]][kMwRowLater]
  uint32_t rows_first, rows_later[8];
  int q_first, n;
  int8_t src_read[QMLE_MAX_QUBITS], src_col[QMLE_MAX_QUBITS];
};
__global__ void __launch_bounds__(1024)
k_mw_purity(const MwPurityArgs a, float *__restrict__ pur_out /* [batch][n] by bit position */) {
  __shared__ double red[16];
  const int b = blockIdx.x, p = blockIdx.y;
  const float *fr = a.first + (size_t)b * a.rows_first * kMwRowFirst;
  double cr = 0, ci = 0, z = 0, tot = 0;
  for (uint32_t i = threadIdx.x; i < a.rows_first; i += blockDim.x) {
    const float *row = fr + (size_t)i * kMwRowFirst;
    const float t = row[36];
    tot += t;
    if (p < kMwT) { cr += row[2 * p]; ci += row[2 * p + 1]; z += row[24 + p]; }
    else if (p < kMwT + a.q_first) z += row[37 + p - kMwT];
    else z += ((i >> (p - kMwT - a.q_first)) & 1u) ? -(double)t : (double)t;
  }
  if (p >= kMwT) {
    const int r = a.src_read[p] - 1, col = a.src_col[p];
    const float *lr = a.later[r] + (size_t)b * a.rows_later[r] * kMwRowLater;
    for (uint32_t i = threadIdx.x; i < a.rows_later[r]; i += blockDim.x) {
      cr += lr[(size_t)i * kMwRowLater + 2 * col];
      ci += lr[(size_t)i * kMwRowLater + 2 * col + 1];
    }
  }
  cr = block_sum_d(cr, red);
  ci = block_sum_d(ci, red);
  z = block_sum_d(z, red);
  tot = block_sum_d(tot, red);
  if (threadIdx.x == 0) {
    const double pa = 0.5 * (tot + z), pd = 0.5 * (tot - z);
    pur_out[(size_t)b * a.n + p] = (float)(pa * pa + pd * pd + 2.0 * (cr * cr + ci * ci));
  }
}

__global__ void k_mw_tile_q(const float *__restrict__ pur, int n, int batch,
                            float *__restrict__ out, float *__restrict__ purities) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  double sum = 0.0;
  for (int p = 0; p < n; ++p) {
    const float v = pur[(size_t)b * n + p];
    sum += v;
    if (purities) purities[(size_t)b * n + (n - 1 - p)] = v;  // index by wire
  }
  out[b] = (float)(2.0 * (1.0 - sum / n));
}


// ---------------------------------------------------------------------------
// angle table from device-resident leaves:  table[b][s] = c[s] + sum_t coef[t] * leaf_{arg[t]}[row][idx[t]]
// (gate angles are affine in params / inputs, ansaetze.py:323-371, model.py:804-816); the row
// of leaf k for flattened sample b is (b / div_k) % mod_k -- the cartesian batch of
// model.py:1449-1481 without materialising the repeats.
// ---------------------------------------------------------------------------
struct AngleLeaves {
  const float *ptr[8];
  long long stride[8];  // floats per row
  int div[8], mod[8];
};

__global__ void __launch_bounds__(256)
k_build_angles(AngleLeaves lv, const int *__restrict__ ptr, const int *__restrict__ arg,
               const int *__restrict__ idx, const float *__restrict__ coef,
               const float *__restrict__ cst, const double *__restrict__ period, int n_slots,
               long long batch, long long b_offset, float *__restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch * n_slots) return;
  const long long b = i / n_slots;
  const int s = (int)(i - b * n_slots);
  const long long gb = b + b_offset;
  // fp64 accumulation, then reduction into (-period/2, period/2] (4 pi for a rotation gate, which
  // depends on angle / 2 only): binary / ternary encodings scale inputs by up to 3^(n-1), and a
  // float32 angle of ~1500 rad would carry 1e-4 rad of rounding into the gate matrices
  double acc = (double)cst[s];
  for (int t = ptr[s]; t < ptr[s + 1]; ++t) {
    const int k = arg[t];
    const long long row = (gb / lv.div[k]) % lv.mod[k];
    acc = fma((double)coef[t], (double)lv.ptr[k][row * lv.stride[k] + idx[t]], acc);
  }
  const double per = period ? period[s] : 0.0;
  if (per > 0.0 && (acc > per || acc < -per)) acc -= per * rint(acc / per);
  out[i] = (float)acc;
}


// vec(rho) measurements: rho[i][j] at flat index i * D + j (ket bits first)
__global__ void __launch_bounds__(256)
k_density_probs(const float2 *__restrict__ rho, int n, float *__restrict__ out) {
  const int b = blockIdx.y;
  const uint64_t D = (uint64_t)1 << n;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D) out[(size_t)b * D + i] = rho[((size_t)b << (2 * n)) + i * (D + 1)].x;
}

__global__ void __launch_bounds__(256)
k_density_expval(const float2 *__restrict__ rho, int n, ObsBits obs, int n_obs,
                 float *__restrict__ out) {
  __shared__ double red[16];
  const int b = blockIdx.x, k = blockIdx.y;
  const uint64_t D = (uint64_t)1 << n;
  const float2 *r = rho + ((size_t)b << (2 * n));
  const int p = obs.bits[k];
  double acc = 0.0;
  for (uint64_t i = threadIdx.x; i < D; i += blockDim.x) {
    const float v = r[i * (D + 1)].x;
    acc += ((i >> p) & 1ull) ? -(double)v : (double)v;
  }
  const double tot = block_sum_d(acc, red);
  if (threadIdx.x == 0) out[(size_t)b * n_obs + k] = (float)tot;
}


// ---------------------------------------------------------------------------------------------
// complex128 engine (the reference's `jax_enable_x64` mode, operations.py:12-16): the same
// plan -- lowered operators after 1-qubit merging, matrices built per sample in fp64 -- applied to
// double-precision amplitudes.  One generic operator applier serves two regimes: the whole
// state in one workgroup's LDS (n <= 13: 2^13 x 16 B = 128 KiB) for the entire circuit +
// measurement, and one streaming launch per operator above.  No fusion, no known-zero tricks:
// this is the accuracy mode (parity 1e-10 against the complex128 oracle), used where float32
// rounding is visible in the result -- e.g. the Fourier-coefficient correlation of analytically
// vanishing coefficients (coefficients.py:966-1650, tests/test_coefficients.py:954-983).
// ---------------------------------------------------------------------------------------------
// Complex arithmetic WITHOUT fused multiply-adds: every product and every sum rounds once, like
// the reference's (XLA / NumPy) complex128 einsum.  The accuracy is the same either way; what
// differs is the structure of the 1e-17 rounding residue, which the Fourier-coefficient
// correlation of analytically vanishing coefficients is made of (tests/test_gpu_fcc.py).
#pragma clang fp contract(off)
__device__ __forceinline__ double2 zmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 zfma(double2 a, double2 b, double2 c) {  // a * b + c
  const double2 p = zmul(a, b);
  return make_double2(p.x + c.x, p.y + c.y);
}
#pragma clang fp contract(fast)
struct F64Obs {
  uint32_t mask[QMLE_MAX_QUBITS];  // bit-position masks of the Z (x) Z ... observables
};

// work items [first, first + step, ...) of ONE lowered operator on the amplitudes `s` of one state
__device__ __forceinline__ void f64_apply(double2 *s, int n, const LoweredOp &op, const double *__restrict__ mrow,
                                          const double *__restrict__ consts, const double *__restrict__ ang,
                                          uint64_t first, uint64_t step) {
  const uint64_t D = (uint64_t)1 << n;
  if (op.kind == LK_DIAG_ALL) {
    const double x = ang[op.slot];
    const double *mark = consts + op.mat_off;
    for (uint64_t i = first; i < D; i += step) {
      double sn, cs;
      sincos(-mark[i] * x, &sn, &cs);
      s[i] = zmul(s[i], make_double2(cs, sn));
    }
    return;
  }
  // positions the operator acts on, ascending, and the mask of its control bits
  int pos[4], np = 0;
  uint64_t cmask = 0;
  auto add = [&](int p) {
    int j = np++;
    while (j > 0 && pos[j - 1] > p) { pos[j] = pos[j - 1]; --j; }
    pos[j] = p;
  };
  add(op.t0);
  if (op.kind == LK_4Q) { add(op.t1); add(op.c0); add(op.c1); }
  else {
    if (op.t1 >= 0) add(op.t1);
    if (op.nc >= 1) { add(op.c0); cmask |= (uint64_t)1 << op.c0; }
    if (op.nc >= 2) { add(op.c1); cmask |= (uint64_t)1 << op.c1; }
  }
  const uint64_t count = D >> np;
  if (op.kind == LK_1Q) {
    const double2 m00 = make_double2(mrow[op.mat_off + 0], mrow[op.mat_off + 1]);
    const double2 m01 = make_double2(mrow[op.mat_off + 2], mrow[op.mat_off + 3]);
    const double2 m10 = make_double2(mrow[op.mat_off + 4], mrow[op.mat_off + 5]);
    const double2 m11 = make_double2(mrow[op.mat_off + 6], mrow[op.mat_off + 7]);
    const uint64_t tb = (uint64_t)1 << op.t0;
    for (uint64_t i = first; i < count; i += step) {
      uint64_t idx = i;
      for (int j = 0; j < np; ++j) idx = ins0_64(idx, pos[j]);
      idx |= cmask;
      const double2 a0 = s[idx], a1 = s[idx | tb];
      s[idx] = zfma(m01, a1, zmul(m00, a0));
      s[idx | tb] = zfma(m11, a1, zmul(m10, a0));
    }
  } else if (op.kind == LK_2Q) {  // row = 2 bit[t0] + bit[t1]
    const double *m = mrow + op.mat_off;
    const uint64_t b0 = (uint64_t)1 << op.t0, b1 = (uint64_t)1 << op.t1;
    for (uint64_t i = first; i < count; i += step) {
      uint64_t idx = i;
      for (int j = 0; j < np; ++j) idx = ins0_64(idx, pos[j]);
      idx |= cmask;
      double2 a[4], r[4];
      for (int k = 0; k < 4; ++k) a[k] = s[idx | ((k & 2) ? b0 : 0) | ((k & 1) ? b1 : 0)];
      for (int rr = 0; rr < 4; ++rr) {
        double2 acc = make_double2(0.0, 0.0);
        for (int c = 0; c < 4; ++c) acc = zfma(make_double2(m[2 * (rr * 4 + c)], m[2 * (rr * 4 + c) + 1]), a[c], acc);
        r[rr] = acc;
      }
      for (int k = 0; k < 4; ++k) s[idx | ((k & 2) ? b0 : 0) | ((k & 1) ? b1 : 0)] = r[k];
    }
  } else {  // LK_4Q: 16 x 16 on (t0, t1, c0, c1), row bit 3 = t0 ... bit 0 = c1; batch-constant matrix
    const double *m = consts + op.mat_off;
    const uint64_t bb[4] = {(uint64_t)1 << op.t0, (uint64_t)1 << op.t1, (uint64_t)1 << op.c0, (uint64_t)1 << op.c1};
    for (uint64_t i = first; i < count; i += step) {
      uint64_t idx = i;
      for (int j = 0; j < np; ++j) idx = ins0_64(idx, pos[j]);
      double2 a[16], r[16];
      for (int k = 0; k < 16; ++k)
        a[k] = s[idx | ((k & 8) ? bb[0] : 0) | ((k & 4) ? bb[1] : 0) | ((k & 2) ? bb[2] : 0) | ((k & 1) ? bb[3] : 0)];
      for (int rr = 0; rr < 16; ++rr) {
        double2 acc = make_double2(0.0, 0.0);
        for (int c = 0; c < 16; ++c) acc = zfma(make_double2(m[2 * (rr * 16 + c)], m[2 * (rr * 16 + c) + 1]), a[c], acc);
        r[rr] = acc;
      }
      for (int k = 0; k < 16; ++k)
        s[idx | ((k & 8) ? bb[0] : 0) | ((k & 4) ? bb[1] : 0) | ((k & 2) ? bb[2] : 0) | ((k & 1) ? bb[3] : 0)] = r[k];
    }
  }
}

// whole circuit + measurement of one sample per workgroup, state in LDS (n <= 13)
__global__ void __launch_bounds__(256)
k64_lds(const LoweredOp *__restrict__ ops, int n_ops, int n, const double *__restrict__ mats, uint32_t mat_floats,
        const double *__restrict__ consts, const double *__restrict__ angles, int n_slots, int meas,
        F64Obs obs, int n_obs, void *__restrict__ out) {
  extern __shared__ double2 st64[];
  __shared__ double red[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const uint64_t D = (uint64_t)1 << n;
  for (uint64_t i = tid; i < D; i += blockDim.x) st64[i] = make_double2(i == 0 ? 1.0 : 0.0, 0.0);
  __syncthreads();
  const double *mrow = mats + (size_t)b * mat_floats;
  const double *ang = angles ? angles + (size_t)b * n_slots : nullptr;
  for (int k = 0; k < n_ops; ++k) {
    f64_apply(st64, n, ops[k], mrow, consts, ang, (uint64_t)tid, (uint64_t)blockDim.x);
    __syncthreads();
  }
  if (meas == QMLE_MEAS_STATE || meas == QMLE_MEAS_DENSITY) {
    double2 *o = (double2 *)out + (size_t)b * D;
    for (uint64_t i = tid; i < D; i += blockDim.x) o[i] = st64[i];
  } else if (meas == QMLE_MEAS_PROBS) {
    double *o = (double *)out + (size_t)b * D;
    for (uint64_t i = tid; i < D; i += blockDim.x) o[i] = st64[i].x * st64[i].x + st64[i].y * st64[i].y;
  } else {
    for (int k = 0; k < n_obs; ++k) {
      double acc = 0.0;
      for (uint64_t i = tid; i < D; i += blockDim.x) {
        const double p = st64[i].x * st64[i].x + st64[i].y * st64[i].y;
        acc += (__builtin_popcountll(i & obs.mask[k]) & 1) ? -p : p;
      }
      const double t = block_sum_d(acc, red);
      if (tid == 0) ((double *)out)[(size_t)b * n_obs + k] = t;
      __syncthreads();
    }
  }
}

// n >= 14: states in HBM, one launch per operator
__global__ void __launch_bounds__(256) k64_init(double2 *__restrict__ states, int n) {
  const uint64_t D = (uint64_t)1 << n;
  double2 *s = states + ((size_t)blockIdx.y << n);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < D; i += (uint64_t)gridDim.x * blockDim.x)
    s[i] = make_double2(i == 0 ? 1.0 : 0.0, 0.0);
}
__global__ void __launch_bounds__(256)
k64_op(double2 *__restrict__ states, int n, LoweredOp op, const double *__restrict__ mats, uint32_t mat_floats,
       const double *__restrict__ consts, const double *__restrict__ angles, int n_slots) {
  const int b = blockIdx.y;
  f64_apply(states + ((size_t)b << n), n, op, mats + (size_t)b * mat_floats, consts,
            angles ? angles + (size_t)b * n_slots : nullptr, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x,
            (uint64_t)gridDim.x * blockDim.x);
}
__global__ void __launch_bounds__(256)
k64_probs(const double2 *__restrict__ states, double *__restrict__ out, uint64_t total) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x)
    out[i] = states[i].x * states[i].x + states[i].y * states[i].y;
}
__global__ void __launch_bounds__(256)
k64_expval(const double2 *__restrict__ states, int n, F64Obs obs, int n_obs, double *__restrict__ out) {
  __shared__ double red[16];
  const int b = blockIdx.x, k = blockIdx.y;
  const uint64_t D = (uint64_t)1 << n;
  const double2 *s = states + ((size_t)b << n);
  double acc = 0.0;
  for (uint64_t i = threadIdx.x; i < D; i += blockDim.x) {
    const double p = s[i].x * s[i].x + s[i].y * s[i].y;
    acc += (__builtin_popcountll(i & obs.mask[k]) & 1) ? -p : p;
  }
  const double t = block_sum_d(acc, red);
  if (threadIdx.x == 0) out[(size_t)b * n_obs + k] = t;
}
__global__ void __launch_bounds__(256)
k64_density(const double2 *__restrict__ states, int n, double2 *__restrict__ out) {  // rho = |psi><psi|
  const uint64_t D = (uint64_t)1 << n;
  const double2 *s = states + ((size_t)blockIdx.y << n);
  double2 *o = out + ((size_t)blockIdx.y << (2 * n));
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < D * D; e += (uint64_t)gridDim.x * blockDim.x) {
    const double2 a = s[e >> n], c = s[e & (D - 1)];
    o[e] = make_double2(a.x * c.x + a.y * c.y, a.y * c.x - a.x * c.y);
  }
}

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
static inline bool first_use_on_device(int slot) {
  static std::atomic<bool> done[8][kMaxDevices] = {};
  int dev = -1;
  // a device index beyond the table has no slot of its own: its attributes are simply set on
  // every call (idempotent) instead of sharing -- and trusting -- slot 0's flag
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return true;
  return !done[slot][dev].exchange(true);
}

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

int ensure_device_plan(qmle_plan *p) {
  if (p->dev.blob) {
    // the plan's device image lives on the device of its first run: refuse another one rather
    // than hand kernels a pointer they cannot read
    return p->dev.device == current_device() ? QMLE_OK : QMLE_ERR_UNSUPPORTED;
  }
  p->dev.device = current_device();
  const size_t b_ops = align_up(p->dev_ops.size() * sizeof(LoweredOp) + 16, 256);
  const size_t b_build = align_up(p->build_ops.size() * sizeof(BuildOp) + 16, 256);
  const size_t b_groups = align_up(p->groups.size() * sizeof(BuildGroup) + 16, 256);
  const size_t b_consts = align_up(p->consts.size() * sizeof(float) + 16, 256);
  const size_t b_opg = align_up(p->op_groups.size() * sizeof(OpGroup) + 16, 256);
  const size_t b_ops2 = align_up(p->ops2.size() * sizeof(LoweredOp) + 16, 256);
  const size_t b_grp2 = align_up(p->groups2.size() * sizeof(Group2) + 16, 256);
  const size_t b_tbl2 = align_up(p->tbl2.size() * sizeof(uint32_t) + 16, 256);
  const size_t total = b_ops + b_build + b_groups + b_consts + b_opg + b_ops2 + b_grp2 + b_tbl2;
  char *blob = nullptr;
  HIPCHK(hipMalloc((void **)&blob, total));
  p->dev.blob = blob;
  p->dev.blob_bytes = total;
  p->dev.d_ops = (LoweredOp *)blob;
  p->dev.d_build = (BuildOp *)(blob + b_ops);
  p->dev.d_groups = (BuildGroup *)(blob + b_ops + b_build);
  p->dev.d_consts = (float *)(blob + b_ops + b_build + b_groups);
  p->dev.d_op_groups = (OpGroup *)(blob + b_ops + b_build + b_groups + b_consts);
  char *fast = blob + b_ops + b_build + b_groups + b_consts + b_opg;
  p->dev.d_ops2 = (LoweredOp *)fast;
  p->dev.d_groups2 = (Group2 *)(fast + b_ops2);
  p->dev.d_tbl2 = (uint32_t *)(fast + b_ops2 + b_grp2);
  if (!p->ops2.empty())
    HIPCHK(hipMemcpy(p->dev.d_ops2, p->ops2.data(), p->ops2.size() * sizeof(LoweredOp),
                     hipMemcpyHostToDevice));
  if (!p->groups2.empty())
    HIPCHK(hipMemcpy(p->dev.d_groups2, p->groups2.data(), p->groups2.size() * sizeof(Group2),
                     hipMemcpyHostToDevice));
  if (!p->tbl2.empty())
    HIPCHK(hipMemcpy(p->dev.d_tbl2, p->tbl2.data(), p->tbl2.size() * sizeof(uint32_t),
                     hipMemcpyHostToDevice));
  if (!p->op_groups.empty())
    HIPCHK(hipMemcpy(p->dev.d_op_groups, p->op_groups.data(),
                     p->op_groups.size() * sizeof(OpGroup), hipMemcpyHostToDevice));
  if (!p->dev_ops.empty())
    HIPCHK(hipMemcpy(p->dev.d_ops, p->dev_ops.data(), p->dev_ops.size() * sizeof(LoweredOp),
                     hipMemcpyHostToDevice));
  if (!p->build_ops.empty())
    HIPCHK(hipMemcpy(p->dev.d_build, p->build_ops.data(),
                     p->build_ops.size() * sizeof(BuildOp), hipMemcpyHostToDevice));
  if (!p->groups.empty())
    HIPCHK(hipMemcpy(p->dev.d_groups, p->groups.data(), p->groups.size() * sizeof(BuildGroup),
                     hipMemcpyHostToDevice));
  if (!p->consts.empty())
    HIPCHK(hipMemcpy(p->dev.d_consts, p->consts.data(), p->consts.size() * sizeof(float),
                     hipMemcpyHostToDevice));
  return QMLE_OK;
}

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

static TileArgs fill_tile_args(const qmle_plan *p, const Stage &st, float2 *states,
                               const float *mats, const float *angles, bool init_zero, int meas,
                               void *out, const uint32_t *obs_masks, int n_obs,
                               bool from_zero = false) {
  TileArgs a;
  std::memset(&a, 0, sizeof(a));
  a.states = states;
  a.mats = mats;
  a.angles = angles;
  a.consts = p->dev.d_consts;
  a.ops = p->dev.d_ops;
  a.groups = p->dev.d_op_groups + st.grp_begin;
  a.n_groups = st.grp_end - st.grp_begin;
  a.out = out;
  a.mat_floats = p->mat_floats;
  a.n_ops = st.op_end - st.op_begin;
  a.n = p->n;
  a.T = st.T;
  a.L = st.L;
  a.n_slots = p->n_slots;
  a.init_zero = init_zero ? 1 : 0;
  a.meas = meas;
  a.n_obs = n_obs;
  std::memcpy(a.tile_bits, st.tile_bits, sizeof(a.tile_bits));
  std::memcpy(a.outer_bits, st.outer_bits, sizeof(a.outer_bits));
  if (obs_masks) std::memcpy(a.obs_mask, obs_masks, (size_t)n_obs * sizeof(uint32_t));
  a.op_begin = st.op_begin;
  for (int q = 0; q <= QMLE_MAX_QUBITS; ++q) a.qsrc[q] = 64;
  a.qsrc[QMLE_MAX_QUBITS] = 10;
  if (st.T >= 10) {  // element e = tid + it * 2^(T-4): bits 0..5 lane, 6..T-5 wave, T-4..T-1 iteration
    const int tb = st.T - 4;
    for (int j = 0; j < st.T; ++j)
      a.qsrc[(int)st.tile_bits[j]] = (uint8_t)(j < 6 ? j : j < tb ? 16 + (j - 6) : 6 + (j - tb));
    for (int i = 0; i < p->n - st.T; ++i) a.qsrc[(int)st.outer_bits[i]] = (uint8_t)(32 + i);
  }
  if (from_zero && st.zero_in && !init_zero) {
    for (int j = 0; j < st.T; ++j)
      if (st.zero_in & (1u << st.tile_bits[j])) a.zin_local |= 1u << j;
    for (int i = 0; i < p->n - st.T; ++i)
      if (st.zero_in & (1u << st.outer_bits[i])) a.zin_outer |= 1u << i;
  }
  return a;
}

// A run that starts from |0..0> keeps track of the amplitudes that are still exactly zero
// (Stage::zero_in); the prefetching experiment does not.
static bool plan_sparse(const qmle_plan *p) {
  static const bool pf_env = std::getenv("QMLE_PREFETCH") != nullptr;
  return !(p->flags & (QMLE_PLAN_NO_SPARSE | QMLE_PLAN_PREFETCH)) && !pf_env;
}

int launch_tile(const qmle_plan *p, const Stage &st, float2 *states, const float *mats,
                const float *angles, int batch, bool init_zero, int meas, void *out,
                const uint32_t *obs_masks, int n_obs, hipStream_t stream,
                bool from_zero = false, float2 *cols = nullptr, int *row_shift = nullptr) {
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
  const size_t lds = tile_lds_bytes(st.T, st.L, a.slots_in_lds ? a.n_ops : 0);
  if (first_use_on_device(0)) {
    QMLE_LDS_BASE_CHECK(k_tile<false>);
    QMLE_LDS_BASE_CHECK(k_tile<true>);
    HIPCHK(hipFuncSetAttribute((const void *)k_tile<false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_tile<true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  bool has_dense4 = false;  // 16x16 Kraus superoperators: separate instantiation, so that the
                            // common kernel keeps its register budget
  for (int g = st.grp_begin; g < st.grp_end; ++g) has_dense4 |= p->op_groups[g].kind == GK_DENSE4;
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
    if (first_use_on_device(1)) {
      QMLE_LDS_BASE_CHECK(k_tile_pf<false>);
      QMLE_LDS_BASE_CHECK(k_tile_pf<true>);
      HIPCHK(hipFuncSetAttribute((const void *)k_tile_pf<false>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void *)k_tile_pf<true>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
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
    const uint64_t count = ((uint64_t)batch << p->n) / 2u;  // float4 = two amplitudes
    for (uint64_t done = 0; done < count;) {  // (grid.x < 2^31 workgroups per launch)
      const uint64_t part = std::min<uint64_t>(count - done, (uint64_t)1 << 38);
      hipLaunchKernelGGL(k_fill_zero, dim3((unsigned)((part + 255u) / 256u)), dim3(256), 0, stream,
                         reinterpret_cast<float4 *>(states) + done, part);
      done += part;
    }
    a.compact = 1;  // grid = the tiles that can be non-zero = tile 0
    a.tile_free = 0u;
    grid.x = 1u;
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
    if (first_use_on_device(2)) {
#define QMLE_T2_LDS(NT, ME, MU)                                                   \
  QMLE_LDS_BASE_CHECK((k_tile2<NT, ME, MU>));                                      \
  HIPCHK(hipFuncSetAttribute((const void *)k_tile2<NT, ME, MU>,                    \
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
      QMLE_T2_LDS(false, false, false); QMLE_T2_LDS(true, false, false);
      QMLE_T2_LDS(false, true, false); QMLE_T2_LDS(true, true, false);
      QMLE_T2_LDS(false, false, true); QMLE_T2_LDS(true, false, true);
      QMLE_T2_LDS(false, true, true); QMLE_T2_LDS(true, true, true);
#undef QMLE_T2_LDS
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
    const int tpw_max = tpw_env > 0 ? tpw_env : meas == TM_EXPVAL_PARTIAL ? 8 : 4;
    f.tpw = 1;
    f.tile_stride = 0;
    // (known zeros inside the tile are fine -- the walk's loads skip them; known-zero TILES are not)
    const bool multi_zin = std::getenv("QMLE_NO_MULTI_ZIN") == nullptr;  // (read per launch: the A/B test toggles it)
    if (!a.init_zero && (!a.zin_local || multi_zin) && !a.zin_outer && !a.compact && st.T < p->n &&
        (meas == TM_STORE || meas == TM_PROBS || meas == TM_EXPVAL_PARTIAL)) {
      // (consecutive tile indices differ in the lowest run of outer bit positions only)
      int run0 = 1;
      while (run0 < p->n - st.T && st.outer_bits[run0] == st.outer_bits[0] + run0) ++run0;
      f.tile_stride = 1u << st.outer_bits[0];
      while (f.tpw * 2 <= tpw_max && f.tpw * 2 <= (1 << run0) && grid.x % 2u == 0 &&
             (uint64_t)(grid.x / 2u) * grid.y >= 5120) {
        f.tpw *= 2;
        grid.x /= 2u;
      }
    }
    if (meas == TM_EXPVAL_PARTIAL && f.tpw > 1) {
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
    const size_t lds2 = ((size_t)8 << st.T) +
                        (meas == TM_EXPVAL ? (size_t)QMLE_MAX_QUBITS * (threads >= kWave ? threads / kWave : 1) * sizeof(float) : 0);
    const bool measure = !(meas == TM_STORE || meas == TM_PROBS);
#define QMLE_T2_GO(NT, ME, MU) \
  hipLaunchKernelGGL((k_tile2<NT, ME, MU>), grid, dim3(threads), lds2, stream, a, f)
    const bool multi = f.tpw > 1;
    if (measure) {
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
  if (has_dense4) hipLaunchKernelGGL(k_tile<true>, grid, dim3(threads), lds, stream, a);
  else hipLaunchKernelGGL(k_tile<false>, grid, dim3(threads), lds, stream, a);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

// k_reg_measure* takes the last pass of a <Z> run when all its gates share one register-tile
// group (expval_kernel_of, qmle_plan.cpp)
static int reg_measure_kind(const qmle_plan *p, size_t si, int n_obs) {
  static const bool off = std::getenv("QMLE_NO_REG_MEASURE") != nullptr;
  if (off || n_obs < 1 || n_obs > 32) return 0;
  return expval_kernel_of(p, si, plan_sparse(p));
}

static int launch_reg_measure(const qmle_plan *p, const Stage &st, int kind, float2 *states,
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

template <int MODE>
void launch_direct_mode(bool diag, bool nt, dim3 grid, hipStream_t stream, float4 *st, int n,
                        int pt, int pc, const float *mats, uint32_t mat_floats,
                        uint32_t mat_off, uint64_t items) {
#define QMLE_LAUNCH_DIRECT(D, N)                                                              \
  hipLaunchKernelGGL((k_direct_1q<MODE, D, N>), grid, dim3(256), 0, stream, st, n, pt, pc, mats, \
                     mat_floats, mat_off, items)
  if (diag) { if (nt) QMLE_LAUNCH_DIRECT(true, true); else QMLE_LAUNCH_DIRECT(true, false); }
  else { if (nt) QMLE_LAUNCH_DIRECT(false, true); else QMLE_LAUNCH_DIRECT(false, false); }
#undef QMLE_LAUNCH_DIRECT
}

int launch_direct(const qmle_plan *p, const LoweredOp &op, float2 *states, const float *mats,
                  int batch, hipStream_t stream) {
  const int n = p->n;
  const bool diag = op.flags & LF_DIAG;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  const int pt = op.t0, pc = op.c0;
  int mode;
  uint64_t items;
  static const bool k1_plain = std::getenv("QMLE_K1_PLAIN") != nullptr;
  if (op.nc == 0) {
    if (pt >= 1) {
      mode = 0;
      items = diag ? chunks : chunks >> 1;
      // dense gate, state >= 2^12 chunks: lane exchange for bits 1..6, 4-row bursts for bits >= 21
      if (!diag && !k1_plain && n >= 14) {
        if (pt <= 6) { mode = 5; items = chunks >> 1; }
        else if (pt >= 21) { mode = 6; items = chunks >> 3; }
      }
    } else { mode = 1; items = chunks; }
  } else {
    if (pc >= 1 && pt >= 1) { mode = 2; items = diag ? chunks >> 1 : chunks >> 2; }
    else if (pt == 0) { mode = 3; items = chunks >> 1; }
    else { mode = 4; items = diag ? chunks : chunks >> 1; }
    // control and target both inside a wave's 1 KiB: one contiguous float4 per lane (mode 7)
    // (controls on bits >= 4 select whole 128-byte lines: mode 2 moves half the state, 0.37 vs 0.65 ms)
    if (!diag && !k1_plain && n >= 14 && pt >= 1 && pt <= 6 && pc >= 0 && pc <= 3) { mode = 7; items = chunks; }
  }
  if (items == 0) items = 1;
  // streaming (non-temporal) accesses once the working set dwarfs the Infinity Cache
  const bool nt = ((size_t)batch << n) * sizeof(float2) >= ((size_t)1 << 30);
  dim3 grid((unsigned)((items + 255) / 256), (unsigned)batch);
  float4 *st = reinterpret_cast<float4 *>(states);
  switch (mode) {
    case 0: launch_direct_mode<0>(diag, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items); break;
    case 1: launch_direct_mode<1>(diag, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items); break;
    case 2: launch_direct_mode<2>(diag, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items); break;
    case 3: launch_direct_mode<3>(diag, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items); break;
    case 5: launch_direct_mode<5>(false, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items); break;
    case 6: launch_direct_mode<6>(false, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items); break;
    case 7: launch_direct_mode<7>(false, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items); break;
    default: launch_direct_mode<4>(diag, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items); break;
  }
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

struct ProfScope {  // records a start/stop event pair around one stage launch
  qmle_plan *p;
  hipStream_t stream;
  size_t slot;
  bool active;
  ProfScope(qmle_plan *plan, int stage_idx, hipStream_t s) : p(plan), stream(s), slot(0), active(false) {
    StageProfile &pr = plan->prof;
    if (pr.on && pr.used < pr.start.size()) {
      slot = pr.used++;
      pr.stage[slot] = stage_idx;
      active = hipEventRecord((hipEvent_t)pr.start[slot], stream) == hipSuccess;
    }
  }
  ~ProfScope() {
    if (active) (void)hipEventRecord((hipEvent_t)p->prof.stop[slot], stream);
  }
};

int expval_blocks(int n) {
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  const uint64_t seg = (uint64_t)kEzThreads * kEzUnroll;
  uint64_t n_seg = (chunks + seg - 1) / seg;
  if (n_seg > 2048) n_seg = 2048;
  return (int)n_seg;
}

int run_expval(const float2 *states, int n, int batch, const int8_t *obs_bits, int n_obs,
               float *d_out, void *ws, size_t ws_bytes, hipStream_t stream) {
  if (n_obs < 1 || n_obs > QMLE_MAX_QUBITS) return QMLE_ERR_INVALID_ARG;
  const int nb = expval_blocks(n);
  const size_t need = (size_t)batch * nb * (QMLE_MAX_QUBITS + 1) * sizeof(float);
  if (ws_bytes < need) return QMLE_ERR_WORKSPACE;
  ObsBits ob;
  for (int k = 0; k < QMLE_MAX_QUBITS; ++k) ob.row_mask[k] = 0u;
  for (int k = 0; k < n_obs; ++k) {
    if (obs_bits[k] < 0 || obs_bits[k] >= n) return QMLE_ERR_WIRE_RANGE;
    ob.bits[k] = obs_bits[k];
  }
  hipLaunchKernelGGL(k_expval_partial, dim3(nb, batch), dim3(kEzThreads), 0, stream,
                     reinterpret_cast<const float4 *>(states), n, (float *)ws);
  hipLaunchKernelGGL(k_expval_final, dim3(batch, n_obs), dim3(256), 0, stream,
                     (const float *)ws, nb, n_obs, ob, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int qmle_sv_version(void) { return QMLE_SV_VERSION; }

const char *qmle_status_string(int status) {
  switch (status) {
    case QMLE_OK: return "ok";
    case QMLE_ERR_INVALID_ARG: return "invalid argument";
    case QMLE_ERR_WIRE_COUNT: return "wrong number of wires for gate";
    case QMLE_ERR_DUPLICATE_WIRES: return "duplicate wires";
    case QMLE_ERR_WIRE_RANGE: return "wire index out of range";
    case QMLE_ERR_UNKNOWN_OP: return "unknown opcode";
    case QMLE_ERR_MEAS_TYPE: return "unknown measurement type";
    case QMLE_ERR_WORKSPACE: return "workspace too small";
    case QMLE_ERR_HIP: return "HIP runtime error";
    case QMLE_ERR_NO_DEVICE: return "no HIP device";
    case QMLE_ERR_UNSUPPORTED: return "unsupported configuration";
    case QMLE_ERR_SLOT_RANGE: return "angle slot out of range";
    case QMLE_ERR_INTERNAL: return "internal invariant violated (kernel LDS layout)";
    default: return "unknown status";
  }
}

int qmle_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

int qmle_plan_create(const qmle_op *ops, int n_ops, int n_qubits, int n_slots,
                     const float *consts, int n_consts, unsigned flags, qmle_plan **out) {
  if (!out || n_ops < 0 || n_slots < 0 || n_consts < 0 || (n_ops > 0 && !ops))
    return QMLE_ERR_INVALID_ARG;
  *out = nullptr;
  qmle_plan *p = new (std::nothrow) qmle_plan();
  if (!p) return QMLE_ERR_INVALID_ARG;
  flags &= ~QMLE_PLAN_INTERNAL_ZERO_RUN;  // internal: set below on the plans only qmle_run_batch executes
  p->n = n_qubits;
  p->n_slots = n_slots;
  p->flags = flags;
  p->ops.assign(ops, ops + n_ops);
  if (n_consts > 0) p->consts.assign(consts, consts + n_consts);
  p->n_user_consts = (size_t)(n_consts > 0 ? n_consts : 0);
  const int rc = compile_plan(p);
  if (rc != QMLE_OK) {
    delete p;
    return rc;
  }
  // The same tape scheduled for runs from |0..0> only (wider first tile): what qmle_run_batch
  // executes in place of `p` when the pass-cost model prefers it.  qmle_apply_inplace and the
  // adjoint sweep apply stages to LIVE states and keep `p`'s own schedule.
  static const bool no_wide = std::getenv("QMLE_NO_WIDE_FIRST") != nullptr;
  if (!no_wide && !p->whole_state_lds && !(flags & (QMLE_PLAN_NO_FUSION | QMLE_PLAN_PREFETCH)) &&
      !((flags >> 8) & 0xffffu) && p->stages.size() >= 2 && p->stages[0].kind == ST_TILE) {
    qmle_plan *v = new (std::nothrow) qmle_plan();
    if (v) {
      v->n = n_qubits;
      v->n_slots = n_slots;
      v->flags = flags | QMLE_PLAN_INTERNAL_ZERO_RUN;
      v->ops = p->ops;
      v->consts.assign(p->consts.begin(), p->consts.begin() + (n_consts > 0 ? n_consts : 0));
      const bool forced = std::getenv("QMLE_FORCE_CAND") != nullptr;  // (tuning: always run the forced schedule)
      if (compile_plan(v) == QMLE_OK && v->mat_floats == p->mat_floats && (forced || v->model_cost < p->model_cost - 0.5))
        p->zero_variant = v;
      else
        delete v;
    }
  }
  // <Z> measurements run a second plan without the trailing gates that only relabel basis
  // states or add phases (they are folded into the observables at run time)
  if (!(flags & (QMLE_PLAN_NO_ABSORB | QMLE_PLAN_NO_FUSION))) {
    std::vector<qmle_op> kept;
    split_expval_tail(p->ops, p->n, kept, p->absorbed);
    if (!p->absorbed.empty()) {
      qmle_plan *c = new (std::nothrow) qmle_plan();
      if (c) {
        c->n = n_qubits;
        c->n_slots = n_slots;
        c->flags = flags | QMLE_PLAN_NO_ABSORB | QMLE_PLAN_INTERNAL_ZERO_RUN;  // a child only ever runs from |0..0>
        c->ops = kept;
        c->consts.assign(p->consts.begin(), p->consts.begin() + (n_consts > 0 ? n_consts : 0));
        for (const qmle_op &o : p->absorbed) p->absorbed_algo_bytes += algo_bytes(o, p->n);
        c->extra_algo_last_stage = p->absorbed_algo_bytes;
        if (compile_plan(c) == QMLE_OK) p->expval_child = c;
        else delete c;
        // Folding is not free any more (round 2): a folded CX tail turns <Z_w> into parities,
        // which the last tile pass measures with the general-mask epilogue (per tile: full
        // Walsh-Hadamard transform across the lanes) or, one-group passes on live input, with
        // k_reg_measure -- while the fast tile path applies X / CX for nothing (LDS layout) and
        // then takes the single-bit epilogue, whose sums stay in registers across a workgroup's
        // tiles.  Measured (MI355X, HE circuits, us per state, folded vs applied): n = 24: 2
        // layers 61.9 vs 45.5, 4 layers 115.8 vs 135.0; n = 22, 3 layers 25.4 vs 20.8; n = 20, 4
        // layers 8.9 vs 11.2.  The pass-cost model plus 17 (general-mask epilogue) resp. 12
        // (k_reg_measure on live input), in its units of 36 per read+write pass, picks the faster
        // plan in all of them; plans whose folded form ends in a known-zero special kernel keep it.
        static const bool always_fold = std::getenv("QMLE_ALWAYS_FOLD") != nullptr;
        if (p->expval_child && !always_fold && !c->stages.empty() && !c->whole_state_lds) {
          bool parity = false;
          for (int w = 0; w < p->n; ++w) {
            const uint32_t m = pull_back_z(p->absorbed, w);
            if (m & (m - 1u)) parity = true;
          }
          const size_t last = c->stages.size() - 1;
          const Stage &ls = c->stages[last];
          const int kind = expval_kernel_of(c, last, plan_sparse(c));
          double penalty = 0.0;
          if (ls.kind == ST_TILE && parity) {
            if (kind == 0) penalty = 17.0;
            else if (kind == 1 && (!plan_sparse(c) || ls.zero_in == 0)) penalty = 12.0;
          }
          const double applied_cost = p->zero_variant ? p->zero_variant->model_cost : p->model_cost;
          if (penalty > 0.0 && c->model_cost + penalty > applied_cost) {
            delete c;
            p->expval_child = nullptr;
          }
        }
      }
    }
    if (!p->expval_child) p->absorbed.clear();
  }
  *out = p;
  return QMLE_OK;
}
// (the plan QMLE_MEAS_EXPVAL_Z executes: the child when trailing gates were folded into the
// observables, else the from-|0..0> variant of the plan when there is one)
qmle_plan *qmle_plan_expval_child(qmle_plan *plan) {
  return !plan ? nullptr : plan->expval_child ? plan->expval_child : plan->zero_variant;
}

int qmle_plan_destroy(qmle_plan *plan) {
  if (!plan) return QMLE_OK;
  if (plan->expval_child) (void)qmle_plan_destroy(plan->expval_child);
  if (plan->zero_variant) (void)qmle_plan_destroy(plan->zero_variant);
  if (plan->adj_blob) (void)hipFree(plan->adj_blob);
  if (plan->adjf_blob) (void)hipFree(plan->adjf_blob);
  if (plan->f64_blob) (void)hipFree(plan->f64_blob);
  if (plan->dev.blob) (void)hipFree(plan->dev.blob);
  delete plan;
  return QMLE_OK;
}

int qmle_plan_describe(const qmle_plan *plan, char *buf, size_t cap) {
  if (!plan) return QMLE_ERR_INVALID_ARG;
  const std::string s = describe_plan(plan);
  if (buf && cap > 0) {
    const size_t nc = s.size() < cap - 1 ? s.size() : cap - 1;
    std::memcpy(buf, s.data(), nc);
    buf[nc] = 0;
  }
  return (int)s.size();
}

int qmle_plan_stats(const qmle_plan *plan, int64_t stats[8]) {
  if (!plan || !stats) return QMLE_ERR_INVALID_ARG;
  int direct = 0;
  for (const Stage &s : plan->stages) direct += s.kind == ST_DIRECT;
  stats[0] = (int64_t)plan->ops.size();
  stats[1] = (int64_t)plan->stages.size();
  stats[2] = plan->whole_state_lds ? 1 : 0;
  stats[3] = plan->tile_T;
  stats[4] = plan->mat_floats;
  stats[5] = direct;
  stats[6] = (int64_t)plan->lowered.size();
  stats[7] = (int64_t)plan->algo_bytes_per_state;
  return QMLE_OK;
}

// workspace layout: [matrices: batch * mat_floats] [states: S * D (if needed)]
//                   [expval partials]
static size_t ws_matrix_bytes(const qmle_plan *p, int batch) {
  return align_up((size_t)batch * (p->mat_floats ? p->mat_floats : 1) * sizeof(float), 256);
}
// per-sample gate matrices, then the product stages' group columns (k_fold_columns)
static size_t ws_mats_bytes(const qmle_plan *p, int batch) {
  return ws_matrix_bytes(p, batch) +
         align_up((size_t)batch * (size_t)p->fold_groups * 16 * sizeof(float2), 256) +
         align_up((size_t)batch * 32 * sizeof(float), 256);  // k_mono_coef
}

static int default_states_in_flight(const qmle_plan *p, int batch) {
  // states per launch: the tile passes are LDS/VALU-bound, so big launches (fewer tails)
  // beat Infinity-Cache residency -- measured 3.4k -> 4.1k statevectors/s at n = 24 going
  // from 1 to 32 states in flight, +1 % more at 128 (profiles/r01_in_flight_sweep.txt); runs
  // that skip known zeros are launch-bound at 32 (11.5 M -> 14.3 M -> 15.2 M gate-applies/s at
  // 32 / 128 / 512 states, K2).  32 GiB of state buffers = 256 states at n = 24.
  // Round 2: a plan whose every pass streams the whole state (no known zeros left to skip) is
  // not launch-bound, and its passes run faster on a 4 GiB than on a 32 GiB working set -- K2
  // all-live at n = 24: 111.0 / 111.6 / 108.3 / 107.2 / 107.7 ms per 1024 states for 32 / 16 /
  // 8 / 4 / 2 GiB per launch (the read+write pass: 56.6 vs 51.6 us per state at 256 vs 64
  // states); the known-zero plans keep 32 GiB (3.0 vs 4.9 ms per step at 4 GiB).
  const size_t sb = (size_t)8 << p->n;
  static const long env_mib = [] {
    const char *e = getenv("QMLE_IN_FLIGHT_MIB");  // tuning knob; default from measurements
    return e ? atol(e) : 0L;
  }();
  bool whole_state_every_pass = p->stages.size() >= 2;
  if (plan_sparse(p))
    for (size_t si = 1; si < p->stages.size(); ++si)
      if (p->stages[si].zero_in != 0) whole_state_every_pass = false;
  const size_t budget_mib = env_mib > 0 ? (size_t)env_mib : whole_state_every_pass ? 4096 : 32768;
  size_t s = (budget_mib << 20) / sb;
  if (s < 1) s = 1;
  if (s > (size_t)batch) s = (size_t)batch;
  return (int)s;
}

static int overlap_blocks(int n);
static size_t expval_partial_rows(const qmle_plan *p) {
  size_t rows = (size_t)expval_blocks(p->n);
  if ((size_t)overlap_blocks(p->n) > rows) rows = (size_t)overlap_blocks(p->n);
  if (!p->stages.empty() && p->stages.back().kind == ST_TILE && !p->whole_state_lds) {
    const size_t tiles = (size_t)1 << (p->n - p->stages.back().T);
    if (tiles > rows) rows = tiles;
  }
  return rows;
}

static size_t per_state_ws_bytes(const qmle_plan *p, int meas_type) {
  size_t b = align_up((size_t)8 << p->n, 256);
  if (meas_type == QMLE_MEAS_EXPVAL_Z)
    b += align_up(expval_partial_rows(p) * (QMLE_MAX_QUBITS + 1) * sizeof(float), 256);
  return b;
}

static size_t workspace_bytes_one(const qmle_plan *plan, int batch, int meas_type,
                                  int states_in_flight) {
  size_t total = ws_mats_bytes(plan, batch) + 512;  // + alignment slack
  const bool lds_direct_meas =
      plan->whole_state_lds && (meas_type == QMLE_MEAS_PROBS || meas_type == QMLE_MEAS_EXPVAL_Z);
  if (meas_type != QMLE_MEAS_STATE && !lds_direct_meas) {
    int s = states_in_flight > 0 ? states_in_flight : default_states_in_flight(plan, batch);
    if (s > batch) s = batch;
    total += (size_t)s * per_state_ws_bytes(plan, meas_type);
  }
  return total;
}

size_t qmle_workspace_bytes(const qmle_plan *plan, int batch, int meas_type, int n_obs,
                            int states_in_flight) {
  (void)n_obs;
  if (!plan || batch < 1) return 0;
  size_t total = workspace_bytes_one(plan, batch, meas_type, states_in_flight);
  if (plan->zero_variant) {
    const size_t c = workspace_bytes_one(plan->zero_variant, batch, meas_type, states_in_flight);
    if (c > total) total = c;
  }
  if (meas_type == QMLE_MEAS_EXPVAL_Z && plan->expval_child) {
    const size_t c = workspace_bytes_one(plan->expval_child, batch, meas_type, states_in_flight);
    if (c > total) total = c;
  }
  return total;
}

// Z-parity observables (bit-position masks) of resident states: the stand-alone kernels
static int run_parity_pos(const float2 *states, int n, int batch, const uint32_t *pos_masks,
                          int n_obs, float *d_out, void *ws, size_t ws_bytes,
                          hipStream_t stream) {
  const int nb = overlap_blocks(n);
  if (ws_bytes < (size_t)batch * nb * 8 * sizeof(float)) return QMLE_ERR_WORKSPACE;
  for (int o0 = 0; o0 < n_obs; o0 += 8) {
    ParityMasks pm;
    pm.count = n_obs - o0 < 8 ? n_obs - o0 : 8;
    for (int k = 0; k < 8; ++k) pm.m[k] = k < pm.count ? pos_masks[o0 + k] : 0u;
    hipLaunchKernelGGL(k_parity_partial, dim3(nb, batch), dim3(256), 0, stream,
                       (const float4 *)states, n, pm, (float *)ws);
    hipLaunchKernelGGL(k_parity_final, dim3(batch), dim3(nb >= 256 ? 256 : 64), 0, stream,
                       (const float *)ws, nb, pm.count, n_obs, o0, d_out);
  }
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

static int run_batch_masks(qmle_plan *plan, const float *d_angles, int batch, int meas_type,
                           const uint32_t *obs_masks, int n_obs, void *d_out, void *d_workspace,
                           size_t workspace_bytes, hipStream_t stream);

// <Z..Z> on wire masks (bit w = wire w), pulled back through the folded tail -> position masks
static int build_obs_masks(qmle_plan *plan, qmle_plan **exec, const uint32_t *wire_masks, int n_obs,
                           uint32_t *masks) {
  const int n = plan->n;
  *exec = plan->expval_child ? plan->expval_child : plan;
  for (int k = 0; k < n_obs; ++k) {
    const uint32_t in = wire_masks[k];
    if (in == 0 || (n < 32 && (in >> n))) return QMLE_ERR_WIRE_RANGE;
    uint32_t wm = 0;  // a product of Z's pulls back to the XOR of the factors' pull-backs
    for (int w = 0; w < n; ++w)
      if (in & (1u << w)) wm ^= *exec == plan ? 1u << w : pull_back_z(plan->absorbed, w);
    uint32_t pm = 0;
    for (int w = 0; w < n; ++w)
      if (wm & (1u << w)) pm |= 1u << (n - 1 - w);
    masks[k] = pm;  // (never 0: the pull-back is an invertible linear map)
  }
  return QMLE_OK;
}

int qmle_run_batch(qmle_plan *plan, const float *d_angles, int batch, int meas_type,
                   const int32_t *obs_wires, int n_obs, void *d_out, void *d_workspace,
                   size_t workspace_bytes, qmle_stream stream_) {
  if (!plan || batch < 1 || !d_out || !d_workspace) return QMLE_ERR_INVALID_ARG;
  if (meas_type < QMLE_MEAS_STATE || meas_type > QMLE_MEAS_DENSITY) return QMLE_ERR_MEAS_TYPE;
  if (plan->n_slots > 0 && !d_angles) return QMLE_ERR_INVALID_ARG;
  uint32_t masks[QMLE_MAX_QUBITS];
  qmle_plan *exec = plan;
  if (meas_type == QMLE_MEAS_EXPVAL_Z) {
    if (n_obs < 1 || n_obs > QMLE_MAX_QUBITS || !obs_wires) return QMLE_ERR_INVALID_ARG;
    uint32_t wm[QMLE_MAX_QUBITS];
    for (int k = 0; k < n_obs; ++k) {
      if (obs_wires[k] < 0 || obs_wires[k] >= plan->n) return QMLE_ERR_WIRE_RANGE;
      wm[k] = 1u << obs_wires[k];
    }
    const int rc = build_obs_masks(plan, &exec, wm, n_obs, masks);
    if (rc != QMLE_OK) return rc;
  }
  return run_batch_masks(exec, d_angles, batch, meas_type, masks, n_obs, d_out, d_workspace,
                         workspace_bytes, (hipStream_t)stream_);
}

int qmle_run_batch_parity(qmle_plan *plan, const float *d_angles, int batch,
                          const uint32_t *wire_masks, int n_obs, float *d_out, void *d_workspace,
                          size_t workspace_bytes, qmle_stream stream_) {
  if (!plan || batch < 1 || !d_out || !d_workspace || !wire_masks || n_obs < 1 ||
      n_obs > QMLE_MAX_QUBITS)
    return QMLE_ERR_INVALID_ARG;
  if (plan->n_slots > 0 && !d_angles) return QMLE_ERR_INVALID_ARG;
  uint32_t masks[QMLE_MAX_QUBITS];
  qmle_plan *exec = plan;
  const int rc = build_obs_masks(plan, &exec, wire_masks, n_obs, masks);
  if (rc != QMLE_OK) return rc;
  return run_batch_masks(exec, d_angles, batch, QMLE_MEAS_EXPVAL_Z, masks, n_obs, d_out,
                         d_workspace, workspace_bytes, (hipStream_t)stream_);
}

// Simulate + measure; <Z> observables arrive as bit-position parity masks.
static int run_batch_masks(qmle_plan *plan, const float *d_angles, int batch, int meas_type,
                           const uint32_t *obs_masks, int n_obs, void *d_out, void *d_workspace,
                           size_t workspace_bytes, hipStream_t stream) {
  if (plan->zero_variant) plan = plan->zero_variant;  // every run_batch starts from |0..0>
  const int n = plan->n;
  bool single_bits = true;  // plain Z observables: the 33-sums epilogue serves them all
  int8_t obs_bits[QMLE_MAX_QUBITS];
  if (meas_type == QMLE_MEAS_EXPVAL_Z) {
    for (int k = 0; k < n_obs; ++k) {
      const uint32_t m = obs_masks[k];
      if (m == 0 || (m & (m - 1))) single_bits = false;
      obs_bits[k] = (int8_t)(m ? __builtin_ctz(m) : 0);
    }
  }
  // Parities that touch the LAST tile in at most one position (the rest are outer positions =
  // bits of the tile index) also come out of the 33-sums epilogue: column of that position (or
  // of the total) summed over the tile rows with the sign of the outer part; the general-mask
  // epilogue costs 13 - 17 us per state at n = 24, this one 4 - 6.  (A CX tail that would make
  // every folded parity of an HE ring meet the last tile in one position does not exist: the
  // restrictions of those parities to T wires are T + 1 or T + 2 distinct ranges.)
  bool semi_single = false;
  uint32_t row_masks[QMLE_MAX_QUBITS];
  static const bool no_semi = std::getenv("QMLE_NO_SEMI_SINGLE") != nullptr;
  if (meas_type == QMLE_MEAS_EXPVAL_Z && !single_bits && !no_semi && !plan->stages.empty() &&
      plan->stages.back().kind == ST_TILE && !plan->whole_state_lds) {
    const Stage &ls = plan->stages.back();
    uint32_t tile_mask = 0;
    for (int j = 0; j < ls.T; ++j) tile_mask |= 1u << ls.tile_bits[j];
    semi_single = true;
    for (int k = 0; k < n_obs && semi_single; ++k) {
      const uint32_t m = obs_masks[k], in = m & tile_mask;
      if (m == 0 || (in & (in - 1u))) { semi_single = false; break; }
      obs_bits[k] = (int8_t)(in ? __builtin_ctz(in) : QMLE_MAX_QUBITS);  // column 32: the tile's total
      uint32_t rm = 0;
      for (int i = 0; i < n - ls.T; ++i)
        if ((m >> ls.outer_bits[i]) & 1u) rm |= 1u << i;
      row_masks[k] = rm;
    }
  }
  int rc = ensure_device_plan(plan);
  if (rc != QMLE_OK) return rc;

  char *ws = (char *)d_workspace;
  {
    const size_t mis = (size_t)(256 - ((uintptr_t)ws & 255)) & 255;
    ws += mis;
    if (workspace_bytes < mis) return QMLE_ERR_WORKSPACE;
    workspace_bytes -= mis;
  }
  const size_t mats_b = ws_mats_bytes(plan, batch);
  if (workspace_bytes < mats_b) return QMLE_ERR_WORKSPACE;
  float *d_mats = (float *)ws;
  float2 *d_cols = plan->fold_groups ? (float2 *)(ws + ws_matrix_bytes(plan, batch)) : nullptr;
  float *d_coef = (float *)(ws + ws_matrix_bytes(plan, batch) +
                            align_up((size_t)batch * (size_t)plan->fold_groups * 16 * sizeof(float2), 256));
  ws += mats_b;
  workspace_bytes -= mats_b;

  // per-sample gate matrices for the whole batch (tiny)
  if (!plan->groups.empty()) {
    const int ng = (int)plan->groups.size();
    for (int b0 = 0; b0 < batch; b0 += 65535) {
      const int bc = batch - b0 < 65535 ? batch - b0 : 65535;
      hipLaunchKernelGGL(k_build_matrices, dim3((ng + 63) / 64, bc), dim3(64), 0, stream,
                         plan->dev.d_build, plan->dev.d_groups, ng,
                         d_angles + (size_t)b0 * plan->n_slots, plan->n_slots,
                         plan->dev.d_consts, d_mats + (size_t)b0 * plan->mat_floats,
                         plan->mat_floats);
    }
    HIPCHK(hipGetLastError());
  }

  const size_t D = (size_t)1 << n;
  const size_t sb = D * sizeof(float2);

  // ---- whole state in LDS: one launch does simulate + measure ------------------
  if (plan->whole_state_lds) {
    const Stage &st = plan->stages[0];
    if (meas_type == QMLE_MEAS_STATE || meas_type == QMLE_MEAS_PROBS ||
        meas_type == QMLE_MEAS_EXPVAL_Z) {
      for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int bc = batch - b0 < 65535 ? batch - b0 : 65535;
        const float *mats = d_mats + (size_t)b0 * plan->mat_floats;
        const float *ang = d_angles ? d_angles + (size_t)b0 * plan->n_slots : nullptr;
        ProfScope prof_scope(plan, 0, stream);
        if (meas_type == QMLE_MEAS_STATE)
          rc = launch_tile(plan, st, (float2 *)d_out + (size_t)b0 * D, mats, ang, bc, true,
                           TM_STORE, nullptr, nullptr, 0, stream);
        else if (meas_type == QMLE_MEAS_PROBS)
          rc = launch_tile(plan, st, nullptr, mats, ang, bc, true, TM_PROBS,
                           (float *)d_out + (size_t)b0 * D, nullptr, 0, stream);
        else
          rc = launch_tile(plan, st, nullptr, mats, ang, bc, true, TM_EXPVAL,
                           (float *)d_out + (size_t)b0 * n_obs, obs_masks, n_obs, stream);
        if (rc != QMLE_OK) return rc;
      }
      return QMLE_OK;
    }
  }

  // ---- general path: states resident in HBM, sample-major chunks ----------------
  float2 *d_states;
  int in_flight;
  if (meas_type == QMLE_MEAS_STATE) {
    d_states = (float2 *)d_out;
    in_flight = default_states_in_flight(plan, batch);  // sample-major: stay cache-resident
  } else {
    in_flight = (int)(workspace_bytes / per_state_ws_bytes(plan, meas_type));
    if (in_flight < 1) return QMLE_ERR_WORKSPACE;
    if (in_flight > batch) in_flight = batch;
    const int dflt = default_states_in_flight(plan, batch);
    if (in_flight > dflt) in_flight = dflt;
    d_states = (float2 *)ws;
    ws += (size_t)in_flight * align_up(sb, 256);
  }
  if (in_flight > 65535) in_flight = 65535;
  void *d_partial = ws;
  const size_t partial_bytes =
      (size_t)in_flight * expval_partial_rows(plan) * (QMLE_MAX_QUBITS + 1) * sizeof(float);
  // <Z> straight out of the last tile pass (no store of the final state, no extra read)
  const bool fuse_expval = meas_type == QMLE_MEAS_EXPVAL_Z && !plan->stages.empty() &&
                           plan->stages.back().kind == ST_TILE;

  for (int b0 = 0; b0 < batch; b0 += in_flight) {
    const int bc = batch - b0 < in_flight ? batch - b0 : in_flight;
    float2 *stc = meas_type == QMLE_MEAS_STATE ? d_states + (size_t)b0 * D : d_states;
    const float *mats = d_mats + (size_t)b0 * plan->mat_floats;
    const float *ang = d_angles ? d_angles + (size_t)b0 * plan->n_slots : nullptr;
    bool initialised = false;
    int reg_q = -1;  // >= 0: the last pass ran as k_reg_measure with 2^reg_q tiles per row
    int tile_row_shift = 0;  // k_tile2's multi-tile measuring variant: 2^shift tiles per row
    for (size_t si = 0; si < plan->stages.size(); ++si) {
      const Stage &st = plan->stages[si];
      ProfScope prof_scope(plan, (int)si, stream);
      if (st.kind == ST_TILE) {
        const bool last_fused = fuse_expval && si + 1 == plan->stages.size();
        const int tm = !last_fused ? TM_STORE : (single_bits || semi_single) ? TM_EXPVAL_PARTIAL : TM_EXPVAL_MASKS;
        reg_q = -1;
        int reg_kind = last_fused && initialised ? reg_measure_kind(plan, si, n_obs) : 0;
        // (k_reg_measure on live input is the slowest way to take parities; its known-zero forms
        // -- FOLD, mono -- keep priority)
        if (reg_kind == 1 && semi_single) reg_kind = 0;
        if (reg_kind) {
          rc = launch_reg_measure(plan, st, reg_kind, stc, mats, ang, bc, d_partial, obs_masks,
                                  n_obs, stream, &reg_q, d_coef + (size_t)b0 * 32);
        } else
        rc = launch_tile(plan, st, stc, mats, ang, bc, !initialised, tm,
                         last_fused ? d_partial : nullptr, last_fused ? obs_masks : nullptr,
                         last_fused ? n_obs : 0, stream, /*from_zero=*/true,
                         d_cols ? d_cols + (size_t)b0 * plan->fold_groups * 16 : nullptr,
                         last_fused && single_bits ? &tile_row_shift : nullptr);
        initialised = true;
      } else {
        if (!initialised) {
          hipLaunchKernelGGL(k_init_zero, dim3(grid_for(D / 2, 256), bc), dim3(256), 0, stream,
                             reinterpret_cast<float4 *>(stc), n);
          initialised = true;
        }
        if (st.kind == ST_DIRECT) {
          rc = launch_direct(plan, plan->dev_ops[st.op_begin], stc, mats, bc, stream);
        } else {
          const LoweredOp &o = plan->dev_ops[st.op_begin];
          hipLaunchKernelGGL(k_diag_all, dim3(grid_for(D / 2, 256), bc), dim3(256), 0, stream,
                             reinterpret_cast<float4 *>(stc), n, plan->dev.d_consts + o.mat_off,
                             ang, plan->n_slots, o.slot);
          rc = QMLE_OK;
        }
      }
      if (rc != QMLE_OK) return rc;
    }
    if (!initialised)
      hipLaunchKernelGGL(k_init_zero, dim3(grid_for(D / 2, 256), bc), dim3(256), 0, stream,
                         reinterpret_cast<float4 *>(stc), n);
    // measure this chunk
    if (meas_type == QMLE_MEAS_PROBS) {
      const uint64_t tc = (uint64_t)bc * (D / 2);
      hipLaunchKernelGGL(k_probs, dim3(grid_for(tc, 256)), dim3(256), 0, stream,
                         reinterpret_cast<const float4 *>(stc),
                         reinterpret_cast<float2 *>((float *)d_out + (size_t)b0 * D), tc);
    } else if (meas_type == QMLE_MEAS_EXPVAL_Z && fuse_expval) {
      ObsBits ob;  // column of the 33-float row: the bit's sum, or (masks) the observable's own
      const bool by_position = (single_bits || semi_single) && reg_q < 0;
      for (int k = 0; k < QMLE_MAX_QUBITS; ++k) ob.row_mask[k] = 0u;
      for (int k = 0; k < n_obs; ++k) {
        ob.bits[k] = by_position ? obs_bits[k] : (int8_t)k;
        if (by_position && semi_single) ob.row_mask[k] = row_masks[k];
      }
      const int tiles = (1 << (n - plan->stages.back().T)) >> (reg_q < 0 ? tile_row_shift : reg_q);
      hipLaunchKernelGGL(k_expval_final, dim3(bc, n_obs), dim3(256), 0, stream, (const float *)d_partial,
                         tiles, n_obs, ob, (float *)d_out + (size_t)b0 * n_obs);
    } else if (meas_type == QMLE_MEAS_EXPVAL_Z) {
      rc = single_bits
               ? run_expval(stc, n, bc, obs_bits, n_obs, (float *)d_out + (size_t)b0 * n_obs,
                            d_partial, partial_bytes, stream)
               : run_parity_pos(stc, n, bc, obs_masks, n_obs, (float *)d_out + (size_t)b0 * n_obs,
                                d_partial, partial_bytes, stream);
      if (rc != QMLE_OK) return rc;
    } else if (meas_type == QMLE_MEAS_DENSITY) {
      if (n > 15) return QMLE_ERR_UNSUPPORTED;
      hipLaunchKernelGGL(k_density, dim3(grid_for(D * D, 256, 1u << 20), bc), dim3(256), 0, stream, stc,
                         (float2 *)d_out + (size_t)b0 * D * D, n);
    }
    HIPCHK(hipGetLastError());
  }
  return QMLE_OK;
}

// Apply the plan's passes IN PLACE to resident states (no |0..0> initialisation, no
// measurement): the gate-application hot loop on its own (simulation.py:102-103).
// One stage of a plan applied in place to resident states.
static int run_stage_inplace(qmle_plan *plan, const Stage &st, float2 *d_states, const float *d_mats,
                             const float *d_angles, int batch, hipStream_t stream) {
  const int n = plan->n;
  const size_t D = (size_t)1 << n;
  if (st.kind == ST_TILE)
    return launch_tile(plan, st, d_states, d_mats, d_angles, batch, false, TM_STORE, nullptr,
                       nullptr, 0, stream);
  if (st.kind == ST_DIRECT)
    return launch_direct(plan, plan->dev_ops[st.op_begin], d_states, d_mats, batch, stream);
  const LoweredOp &o = plan->dev_ops[st.op_begin];
  hipLaunchKernelGGL(k_diag_all, dim3(grid_for(D / 2, 256), batch), dim3(256), 0, stream,
                     reinterpret_cast<float4 *>(d_states), n, plan->dev.d_consts + o.mat_off,
                     d_angles, plan->n_slots, o.slot);
  return QMLE_OK;
}

int qmle_apply_inplace(qmle_plan *plan, const float *d_angles, int batch, void *d_states,
                       void *d_workspace, size_t workspace_bytes, qmle_stream stream_) {
  if (!plan || batch < 1 || batch > 65535 || !d_states || !d_workspace) return QMLE_ERR_INVALID_ARG;
  if (plan->n_slots > 0 && !d_angles) return QMLE_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  int rc = ensure_device_plan(plan);
  if (rc != QMLE_OK) return rc;
  char *ws = (char *)d_workspace;
  const size_t mis = (size_t)(256 - ((uintptr_t)ws & 255)) & 255;
  if (workspace_bytes < mis + ws_mats_bytes(plan, batch)) return QMLE_ERR_WORKSPACE;
  float *d_mats = (float *)(ws + mis);
  if (!plan->groups.empty()) {
    const int ng = (int)plan->groups.size();
    hipLaunchKernelGGL(k_build_matrices, dim3((ng + 63) / 64, batch), dim3(64), 0, stream,
                       plan->dev.d_build, plan->dev.d_groups, ng, d_angles, plan->n_slots,
                       plan->dev.d_consts, d_mats, plan->mat_floats);
  }
  int stage_idx = -1;
  for (const Stage &st : plan->stages) {
    ProfScope prof_scope(plan, ++stage_idx, stream);
    rc = run_stage_inplace(plan, st, (float2 *)d_states, d_mats, d_angles, batch, stream);
    if (rc != QMLE_OK) return rc;
  }
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

// ---- adjoint gradient ------------------------------------------------------------------------
static uint32_t wires_to_pos(uint32_t wires, int n) {
  uint32_t m = 0;
  for (int w = 0; w < n; ++w)
    if (wires & (1u << w)) m |= 1u << (n - 1 - w);
  return m;
}
static int adj_blocks(int n) {
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  uint64_t b = (chunks + 256 * 8 - 1) / (256 * 8);
  if (b < 1) b = 1;
  if (b > 1024) b = 1024;
  return (int)b;
}
struct AdjLayout { size_t states, lam, mats, ang2, partial, fwd_ws, lds_ops, lds_terms, lds_fmats, tile_partial, total; };
static AdjLayout adj_layout(const qmle_plan *fwd, const qmle_plan *rev, int batch) {
  AdjLayout L;
  const size_t sb = (size_t)batch * ((size_t)8 << fwd->n);
  L.states = 0;
  L.lam = sb;  // lambda DIRECTLY behind psi: one batch of 2B states for the backward gates
  L.mats = align_up(2 * sb, 256);
  L.ang2 = L.mats + ws_mats_bytes(rev, 2 * batch);
  L.partial = L.ang2 + align_up((size_t)2 * batch * (rev->n_slots ? rev->n_slots : 1) * sizeof(float), 256);
  L.fwd_ws = L.partial + align_up((size_t)batch * adj_blocks(fwd->n) * sizeof(float2), 256);
  L.lds_ops = L.fwd_ws + workspace_bytes_one(fwd, batch, QMLE_MEAS_STATE, 0) + 256;
  L.lds_terms = L.lds_ops + align_up(rev->lowered.size() * sizeof(LoweredOp) + 16, 256);
  L.lds_fmats = L.lds_terms + align_up(rev->lowered.size() * sizeof(AdjTermDev) + 16, 256);
  L.tile_partial = L.lds_fmats + ws_mats_bytes(fwd, batch) + 256;
  size_t tp = 0;  // fused tile passes: [B][tiles][terms of the stage]
  for (const Stage &st : rev->stages)
    if (st.kind == ST_TILE && !rev->whole_state_lds) {
      const size_t need = ((size_t)batch << (rev->n - st.T)) * (size_t)(st.op_end - st.op_begin) * sizeof(float);
      if (need > tp) tp = need;
    }
  L.total = L.tile_partial + align_up(tp, 256) + 256;
  return L;
}
size_t qmle_adjoint_workspace_bytes(const qmle_plan *fwd, const qmle_plan *rev, int batch) {
  if (!fwd || !rev || batch < 1) return 0;
  return adj_layout(fwd, rev, batch).total + 256;
}

int qmle_adjoint_gradient(qmle_plan *fwd, qmle_plan *rev, const float *d_angles_fwd,
                          const float *d_angles_rev, int batch, const float *d_weights,
                          const uint32_t *obs_wire_masks, int n_obs,
                          const qmle_adjoint_term *terms, int n_terms, float *d_grad,
                          int n_grad_slots, void *d_workspace, size_t workspace_bytes,
                          qmle_stream stream_) {
  if (!fwd || !rev || batch < 1 || 2 * batch > 65535 || !d_weights || !obs_wire_masks ||
      n_obs < 1 || n_obs > QMLE_MAX_QUBITS || !terms || !d_grad || n_grad_slots < 1 ||
      !d_workspace || fwd->n != rev->n || n_terms != (int)rev->ops.size())
    return QMLE_ERR_INVALID_ARG;
  if ((fwd->n_slots > 0 && !d_angles_fwd) || (rev->n_slots > 0 && !d_angles_rev))
    return QMLE_ERR_INVALID_ARG;
  const int n = fwd->n;
  // rev is either a NO_FUSION plan (one streaming pass per gate) or a NO_MERGE plan (fused tile
  // passes); both keep one source gate per lowered operator
  const bool fused = (rev->flags & QMLE_PLAN_NO_MERGE) && !(rev->flags & QMLE_PLAN_NO_FUSION);
  if (!fused)
    for (const Stage &st : rev->stages)
      if (st.src_ops.size() != 1) return QMLE_ERR_INVALID_ARG;
  for (const auto &srcs : rev->lowered_src)
    if (srcs.size() != 1) return QMLE_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  int rc = ensure_device_plan(rev);
  if (rc != QMLE_OK) return rc;
  char *ws = (char *)d_workspace;
  const size_t mis = (size_t)(256 - ((uintptr_t)ws & 255)) & 255;
  const AdjLayout L = adj_layout(fwd, rev, batch);
  if (workspace_bytes < mis + L.total) return QMLE_ERR_WORKSPACE;
  ws += mis;
  for (int k = 0; k < n_obs; ++k)
    if (obs_wire_masks[k] == 0 || (n < 32 && (obs_wire_masks[k] >> n))) return QMLE_ERR_WIRE_RANGE;
  for (int r = 0; r < n_terms; ++r)
    if (terms[r].out_slot >= n_grad_slots) return QMLE_ERR_SLOT_RANGE;

  // ---- n <= 13: psi and lambda both fit in one workgroup's LDS -> a single launch ----------
  static const bool lds_off = std::getenv("QMLE_ADJOINT_NO_LDS") != nullptr;  // A/B switch
  bool lds_ok = !lds_off && fwd->whole_state_lds && n <= 13 && fwd->stages.size() == 1;
  for (const LoweredOp &o : rev->lowered) lds_ok = lds_ok && o.kind != LK_4Q;
  if (lds_ok) {
    rc = ensure_device_plan(fwd);
    if (rc != QMLE_OK) return rc;
    const Stage &fst = fwd->stages[0];
    const int R = (int)rev->lowered.size();
    std::vector<AdjTermDev> tdev((size_t)(R ? R : 1));
    for (int r = 0; r < R; ++r) {
      if (rev->lowered_src[r].size() != 1) return QMLE_ERR_INVALID_ARG;
      const qmle_adjoint_term &t = terms[rev->lowered_src[r][0]];
      AdjTermDev &d = tdev[r];
      d.out_slot = t.out_slot;
      d.xmask = wires_to_pos(t.x_wires, n);
      d.zmask = wires_to_pos(t.z_wires, n);
      d.pmask = wires_to_pos(t.proj_wires, n);
      d.n_y = t.n_y;
      d.coef = t.coef;
      d.marks_off = t.marks_off;
      d.pad = 0;
      if (t.marks_off >= 0 && (size_t)t.marks_off + ((size_t)1 << n) > rev->consts.size())
        return QMLE_ERR_INVALID_ARG;
    }
    // reverse tape + terms live in a blob owned by the reverse plan (uploaded when they change)
    uint64_t hsh = 1469598103934665603ull;
    for (size_t i = 0; i < tdev.size() * sizeof(AdjTermDev); ++i)
      hsh = (hsh ^ ((const unsigned char *)tdev.data())[i]) * 1099511628211ull;
    hsh ^= (uint64_t)R * 0x9E3779B97F4A7C15ull;
    const size_t ops_b = align_up((size_t)(R ? R : 1) * sizeof(LoweredOp), 256);
    if (!rev->adj_blob || rev->adj_hash != hsh) {
      if (rev->adj_blob) (void)hipFree(rev->adj_blob);
      rev->adj_blob = nullptr;
      HIPCHK(hipMalloc(&rev->adj_blob, ops_b + (size_t)(R ? R : 1) * sizeof(AdjTermDev)));
      if (R) {
        HIPCHK(hipMemcpy(rev->adj_blob, rev->lowered.data(), (size_t)R * sizeof(LoweredOp),
                         hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy((char *)rev->adj_blob + ops_b, tdev.data(), (size_t)R * sizeof(AdjTermDev),
                         hipMemcpyHostToDevice));
      }
      rev->adj_hash = hsh;
    }
    LoweredOp *d_rops = (LoweredOp *)rev->adj_blob;
    AdjTermDev *d_terms = (AdjTermDev *)((char *)rev->adj_blob + ops_b);
    float *fmats = (float *)(ws + L.lds_fmats);
    float *rmats = (float *)(ws + L.mats);
    HIPCHK(hipMemsetAsync(d_grad, 0, (size_t)batch * n_grad_slots * sizeof(float), stream));
    if (!fwd->groups.empty()) {
      const int ng = (int)fwd->groups.size();
      hipLaunchKernelGGL(k_build_matrices, dim3((ng + 63) / 64, batch), dim3(64), 0, stream,
                         fwd->dev.d_build, fwd->dev.d_groups, ng, d_angles_fwd, fwd->n_slots,
                         fwd->dev.d_consts, fmats, fwd->mat_floats);
    }
    if (!rev->groups.empty()) {
      const int ng = (int)rev->groups.size();
      hipLaunchKernelGGL(k_build_matrices, dim3((ng + 63) / 64, batch), dim3(64), 0, stream,
                         rev->dev.d_build, rev->dev.d_groups, ng, d_angles_rev, rev->n_slots,
                         rev->dev.d_consts, rmats, rev->mat_floats);
    }
    AdjLdsArgs a;
    a.fwd = fill_tile_args(fwd, fst, nullptr, fmats, d_angles_fwd, true, TM_STORE, nullptr,
                           nullptr, 0);
    const size_t lds_base = ((size_t)16 << n) + 288 * sizeof(float);
    a.fwd.slots_in_lds = lds_base + (size_t)a.fwd.n_ops * sizeof(OpSlot) <= 160 * 1024 ? 1 : 0;
    const size_t lds = lds_base + (a.fwd.slots_in_lds ? (size_t)a.fwd.n_ops * sizeof(OpSlot) : 0);
    a.rev_ops = d_rops;
    a.terms = d_terms;
    a.n_rev = R;
    a.rev_mats = rmats;
    a.rev_mat_floats = rev->mat_floats;
    a.rev_angles = d_angles_rev;
    a.rev_n_slots = rev->n_slots;
    a.rev_consts = rev->dev.d_consts;
    a.weights = d_weights;
    a.n_obs = n_obs;
    for (int k = 0; k < n_obs; ++k) a.zmask[k] = wires_to_pos(obs_wire_masks[k], n);
    a.grad = d_grad;
    a.n_grad_slots = n_grad_slots;
    if (first_use_on_device(3)) {
      HIPCHK(hipFuncSetAttribute((const void *)k_adjoint_lds<false>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void *)k_adjoint_lds<true>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    bool has_dense4 = false;
    for (int g = fst.grp_begin; g < fst.grp_end; ++g) has_dense4 |= fwd->op_groups[g].kind == GK_DENSE4;
    // all loops are strided, so the sweeps may use more threads than the 2^(n-4) register-tile
    // work items of the forward groups: one 64-lane wave per 256 amplitudes, at least 4 waves
    int threads = tile_threads(n);
    if (threads < 256 && n >= 8) threads = 256;
    if (has_dense4) hipLaunchKernelGGL(k_adjoint_lds<true>, dim3(1, batch), dim3(threads), lds, stream, a);
    else hipLaunchKernelGGL(k_adjoint_lds<false>, dim3(1, batch), dim3(threads), lds, stream, a);
    HIPCHK(hipGetLastError());
    return QMLE_OK;
  }
  if (n < 3) return QMLE_ERR_UNSUPPORTED;  // the per-gate streaming kernels move float4 pairs
  float2 *psi = (float2 *)(ws + L.states);
  float2 *lam = (float2 *)(ws + L.lam);
  float *mats = (float *)(ws + L.mats);
  float *ang2 = (float *)(ws + L.ang2);
  float2 *partial = (float2 *)(ws + L.partial);
  const size_t D = (size_t)1 << n;

  // forward: psi = U_N .. U_1 |0>
  rc = run_batch_masks(fwd, d_angles_fwd, batch, QMLE_MEAS_STATE, nullptr, 0, psi, ws + L.fwd_ws,
                       workspace_bytes - mis - L.fwd_ws, stream);
  if (rc != QMLE_OK) return rc;
  // lambda = (sum_k w_k Z..Z_k) psi
  ZSumArgs z;
  z.n_obs = n_obs;
  for (int k = 0; k < n_obs; ++k) z.mask[k] = wires_to_pos(obs_wire_masks[k], n);
  hipLaunchKernelGGL(k_zsum_apply, dim3(grid_for(D / 2, 256, 4096), batch), dim3(256), 0, stream,
                     (const float4 *)psi, (float4 *)lam, n, d_weights, z);
  HIPCHK(hipMemsetAsync(d_grad, 0, (size_t)batch * n_grad_slots * sizeof(float), stream));
  // the backward gates act on [psi; lambda] as one batch of 2B states: duplicate the angles
  if (rev->n_slots > 0) {
    const size_t ab = (size_t)batch * rev->n_slots * sizeof(float);
    HIPCHK(hipMemcpyAsync(ang2, d_angles_rev, ab, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync((char *)ang2 + ab, d_angles_rev, ab, hipMemcpyDeviceToDevice, stream));
  }
  if (!rev->groups.empty()) {
    const int ng = (int)rev->groups.size();
    hipLaunchKernelGGL(k_build_matrices, dim3((ng + 63) / 64, 2 * batch), dim3(64), 0, stream,
                       rev->dev.d_build, rev->dev.d_groups, ng, ang2, rev->n_slots,
                       rev->dev.d_consts, mats, rev->mat_floats);
  }
  const int nb = adj_blocks(n);
  // ---- fused plan: per-dev_op bookkeeping for the tile passes (cached on the reverse plan) ----
  std::vector<int> st_term_begin, st_n_terms;
  const int32_t *d_term_idx = nullptr, *d_gtype = nullptr, *d_slot_of = nullptr;
  const float *d_coef_of = nullptr;
  if (fused) {
    const size_t nd = rev->dev_ops.size();
    std::vector<int32_t> term_idx(nd ? nd : 1, -1), gtype(nd ? nd : 1, 0), slot_of;
    std::vector<float> coef_of;
    for (const Stage &st : rev->stages) {
      st_term_begin.push_back((int)slot_of.size());
      int cnt = 0;
      if (st.kind == ST_TILE) {
        if ((size_t)16 << st.T > (size_t)150 * 1024 || st.L < 1) return QMLE_ERR_UNSUPPORTED;
        for (int g = st.grp_begin; g < st.grp_end; ++g)
          if (rev->op_groups[g].kind != GK_REG4) return QMLE_ERR_UNSUPPORTED;
        for (int k = st.op_begin; k < st.op_end; ++k) {
          const int src = rev->dev_src[k];
          if (src < 0) return QMLE_ERR_INVALID_ARG;
          const qmle_adjoint_term &t = terms[src];
          if (t.out_slot < 0) continue;
          const LoweredOp &o = rev->dev_ops[k];
          // a single-target generator whose projector is exactly the gate's control
          const int nx = __builtin_popcount(t.x_wires), nz = __builtin_popcount(t.z_wires);
          int gt;
          if (t.marks_off >= 0 || nx > 1 || nz > 1 || (nx && nz && t.x_wires != t.z_wires))
            return QMLE_ERR_UNSUPPORTED;
          if (nx && nz) gt = AG_Y; else if (nx) gt = AG_X; else if (nz) gt = AG_Z; else gt = AG_P1;
          if (o.kind != LK_1Q || o.nc > 1) return QMLE_ERR_UNSUPPORTED;
          term_idx[k] = cnt++;
          gtype[k] = gt;
          slot_of.push_back(t.out_slot);
          coef_of.push_back(t.coef);
        }
      }
      st_n_terms.push_back(cnt);
    }
    uint64_t hsh = 1469598103934665603ull;
    auto mix = [&](const void *ptr, size_t bytes) {
      for (size_t i = 0; i < bytes; ++i) hsh = (hsh ^ ((const unsigned char *)ptr)[i]) * 1099511628211ull;
    };
    mix(term_idx.data(), term_idx.size() * 4);
    mix(gtype.data(), gtype.size() * 4);
    mix(slot_of.data(), slot_of.size() * 4);
    mix(coef_of.data(), coef_of.size() * 4);
    const size_t nt_tot = slot_of.size() ? slot_of.size() : 1;
    const size_t o1 = align_up(term_idx.size() * 4, 256), o2 = o1 + align_up(gtype.size() * 4, 256),
                 o3 = o2 + align_up(nt_tot * 4, 256);
    if (!rev->adjf_blob || rev->adjf_hash != hsh) {
      if (rev->adjf_blob) (void)hipFree(rev->adjf_blob);
      rev->adjf_blob = nullptr;
      HIPCHK(hipMalloc(&rev->adjf_blob, o3 + align_up(nt_tot * 4, 256)));
      HIPCHK(hipMemcpy(rev->adjf_blob, term_idx.data(), term_idx.size() * 4, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy((char *)rev->adjf_blob + o1, gtype.data(), gtype.size() * 4, hipMemcpyHostToDevice));
      if (!slot_of.empty()) {
        HIPCHK(hipMemcpy((char *)rev->adjf_blob + o2, slot_of.data(), slot_of.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy((char *)rev->adjf_blob + o3, coef_of.data(), coef_of.size() * 4, hipMemcpyHostToDevice));
      }
      rev->adjf_hash = hsh;
    }
    d_term_idx = (const int32_t *)rev->adjf_blob;
    d_gtype = (const int32_t *)((char *)rev->adjf_blob + o1);
    d_slot_of = (const int32_t *)((char *)rev->adjf_blob + o2);
    d_coef_of = (const float *)((char *)rev->adjf_blob + o3);
    if (first_use_on_device(4)) {
      HIPCHK(hipFuncSetAttribute((const void *)k_tile_adj,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
  }
  float *tile_partial = (float *)(ws + L.tile_partial);
  size_t si = 0;
  for (const Stage &st : rev->stages) {
    const size_t stage_i = si++;
    if (fused && st.kind == ST_TILE) {
      AdjTileArgs A;
      A.t = fill_tile_args(rev, st, psi, mats, ang2, false, TM_STORE, nullptr, nullptr, 0);
      A.t.slots_in_lds = 1;
      A.lam = lam;
      A.term_idx = d_term_idx + st.op_begin;
      A.gtype = d_gtype + st.op_begin;
      A.partial = tile_partial;
      A.n_terms = st_n_terms[stage_i];
      const int threads = 256;
      const size_t lut_n = ((size_t)1 << (st.T - st.L)) < 4 ? 4 : ((size_t)1 << (st.T - st.L));
      const size_t lds = ((size_t)16 << st.T) + 4 * lut_n + (size_t)A.t.n_ops * sizeof(OpSlot) +
                         (size_t)(threads / kWave) * (A.n_terms ? A.n_terms : 1) * sizeof(float);
      if (lds > 160 * 1024) return QMLE_ERR_UNSUPPORTED;
      const unsigned tiles = 1u << (n - st.T);
      hipLaunchKernelGGL(k_tile_adj, dim3(tiles, batch), dim3(threads), lds, stream, A);
      if (A.n_terms)
        hipLaunchKernelGGL(k_adj_tile_final, dim3(batch, A.n_terms), dim3(256), 0, stream,
                           (const float *)tile_partial, (int)tiles, A.n_terms,
                           d_slot_of + st_term_begin[stage_i], d_coef_of + st_term_begin[stage_i],
                           d_grad, n_grad_slots);
      continue;
    }
    const int r = fused ? rev->dev_src[st.op_begin] : st.src_ops[0];
    if (r < 0) return QMLE_ERR_INVALID_ARG;
    const qmle_adjoint_term &t = terms[r];
    if (t.out_slot >= 0) {
      AdjTerm a;
      a.xmask = wires_to_pos(t.x_wires, n);
      a.zmask = wires_to_pos(t.z_wires, n);
      a.pmask = wires_to_pos(t.proj_wires, n);
      a.marks = t.marks_off >= 0 ? rev->dev.d_consts + t.marks_off : nullptr;
      if (t.marks_off >= 0 && (size_t)t.marks_off + D > rev->consts.size()) return QMLE_ERR_INVALID_ARG;
      hipLaunchKernelGGL(k_adj_overlap, dim3(nb, batch), dim3(256), 0, stream, (const float4 *)psi,
                         (const float4 *)lam, n, a, partial);
      hipLaunchKernelGGL(k_adj_final, dim3(batch), dim3(nb >= 256 ? 256 : 64), 0, stream,
                         (const float2 *)partial, nb, t.n_y, t.coef, d_grad, n_grad_slots,
                         t.out_slot);
    }
    rc = run_stage_inplace(rev, st, psi, mats, ang2, 2 * batch, stream);
    if (rc != QMLE_OK) return rc;
  }
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_build_angles(const float *const *d_leaves, const int64_t *leaf_strides,
                      const int32_t *leaf_div, const int32_t *leaf_mod, int n_leaves,
                      const int32_t *d_ptr, const int32_t *d_arg, const int32_t *d_idx,
                      const float *d_coef, const float *d_const, const double *d_period,
                      int n_slots, int64_t batch, int64_t batch_offset, float *d_out,
                      qmle_stream stream) {
  if (n_leaves < 0 || n_leaves > 8 || n_slots < 0 || batch < 1 || !d_out || !d_ptr || !d_const)
    return QMLE_ERR_INVALID_ARG;
  if (n_slots == 0) return QMLE_OK;
  AngleLeaves lv;
  std::memset(&lv, 0, sizeof(lv));
  for (int k = 0; k < n_leaves; ++k) {
    if (!d_leaves[k] || leaf_div[k] < 1 || leaf_mod[k] < 1) return QMLE_ERR_INVALID_ARG;
    lv.ptr[k] = d_leaves[k];
    lv.stride[k] = leaf_strides[k];
    lv.div[k] = leaf_div[k];
    lv.mod[k] = leaf_mod[k];
  }
  const uint64_t total = (uint64_t)batch * (uint64_t)n_slots;
  hipLaunchKernelGGL(k_build_angles, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     lv, d_ptr, d_arg, d_idx, d_coef, d_const, d_period, n_slots, (long long)batch,
                     (long long)batch_offset, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_profile_begin(qmle_plan *plan, int capacity) {
  if (!plan || capacity < 1) return QMLE_ERR_INVALID_ARG;
  StageProfile &pr = plan->prof;
  while ((int)pr.start.size() < capacity) {
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    pr.start.push_back((void *)a);
    pr.stop.push_back((void *)b);
  }
  pr.stage.assign(pr.start.size(), -1);
  pr.used = 0;
  pr.on = true;
  return QMLE_OK;
}

int qmle_profile_end(qmle_plan *plan, double *stage_ms, int64_t *stage_launches, int n_stages) {
  if (!plan || !stage_ms || !stage_launches || n_stages < (int)plan->stages.size())
    return QMLE_ERR_INVALID_ARG;
  StageProfile &pr = plan->prof;
  pr.on = false;
  for (int i = 0; i < n_stages; ++i) { stage_ms[i] = 0.0; stage_launches[i] = 0; }
  for (size_t k = 0; k < pr.used; ++k) {
    HIPCHK(hipEventSynchronize((hipEvent_t)pr.stop[k]));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, (hipEvent_t)pr.start[k], (hipEvent_t)pr.stop[k]));
    stage_ms[pr.stage[k]] += ms;
    stage_launches[pr.stage[k]] += 1;
  }
  const int dropped = pr.used >= pr.start.size() ? 1 : 0;
  for (size_t k = 0; k < pr.start.size(); ++k) {
    (void)hipEventDestroy((hipEvent_t)pr.start[k]);
    (void)hipEventDestroy((hipEvent_t)pr.stop[k]);
  }
  pr.start.clear(); pr.stop.clear(); pr.stage.clear(); pr.used = 0;
  return dropped;  // 1 = the pool filled up (later launches were not timed)
}

size_t qmle_expval_workspace_bytes(int n_qubits, int batch) {
  if (n_qubits < 1 || batch < 1) return 0;
  return (size_t)batch * expval_blocks(n_qubits) * (QMLE_MAX_QUBITS + 1) * sizeof(float) + 256;
}

int qmle_expval_z(const void *d_states, int n_qubits, int batch, const int32_t *obs_wires,
                  int n_obs, float *d_out, void *d_workspace, size_t workspace_bytes,
                  qmle_stream stream) {
  if (!d_states || !d_out || !d_workspace || !obs_wires || n_qubits < 1 ||
      n_qubits > QMLE_MAX_QUBITS || batch < 1 || batch > 65535)
    return QMLE_ERR_INVALID_ARG;
  if (n_obs < 1 || n_obs > QMLE_MAX_QUBITS) return QMLE_ERR_INVALID_ARG;
  int8_t bits[QMLE_MAX_QUBITS];
  for (int k = 0; k < n_obs; ++k) {
    if (obs_wires[k] < 0 || obs_wires[k] >= n_qubits) return QMLE_ERR_WIRE_RANGE;
    bits[k] = (int8_t)(n_qubits - 1 - obs_wires[k]);
  }
  return run_expval((const float2 *)d_states, n_qubits, batch, bits, n_obs, d_out,
                    d_workspace, workspace_bytes, (hipStream_t)stream);
}

int qmle_probs(const void *d_states, int n_qubits, int batch, float *d_out, qmle_stream stream) {
  if (!d_states || !d_out || n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS || batch < 1)
    return QMLE_ERR_INVALID_ARG;
  const uint64_t tc = (uint64_t)batch << (n_qubits - 1);
  hipLaunchKernelGGL(k_probs, dim3(grid_for(tc, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float4 *)d_states, (float2 *)d_out, tc);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_density(const void *d_states, int n_qubits, int batch, void *d_out, qmle_stream stream) {
  if (!d_states || !d_out || n_qubits < 1 || batch < 1 || batch > 65535) return QMLE_ERR_INVALID_ARG;
  if (n_qubits > 15) return QMLE_ERR_UNSUPPORTED;
  const uint64_t D = (uint64_t)1 << n_qubits;
  hipLaunchKernelGGL(k_density, dim3(grid_for(D * D, 256, 1u << 20), batch), dim3(256), 0,
                     (hipStream_t)stream, (const float2 *)d_states, (float2 *)d_out, n_qubits);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_marginal_probs(const void *d_states, int n_qubits, int batch, const int32_t *keep_wires,
                        int n_keep, float *d_out, qmle_stream stream_) {
  if (!d_states || !d_out || !keep_wires || n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS ||
      batch < 1 || batch > 65535 || n_keep < 1 || n_keep > n_qubits || n_keep > 24)
    return QMLE_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  // kept wires stay in ascending wire order regardless of `keep` order
  // (jaqsi.py:141-146): output bit k (LSB first) <- the k-th LARGEST wire.
  uint64_t mask = 0;
  for (int k = 0; k < n_keep; ++k) {
    if (keep_wires[k] < 0 || keep_wires[k] >= n_qubits) return QMLE_ERR_WIRE_RANGE;
    if (mask & (1ull << keep_wires[k])) return QMLE_ERR_DUPLICATE_WIRES;
    mask |= 1ull << keep_wires[k];
  }
  KeepBits kb;
  kb.n_keep = n_keep;
  int k = 0;
  for (int w = n_qubits - 1; w >= 0; --w)
    if (mask & (1ull << w)) kb.bits[k++] = (int8_t)(n_qubits - 1 - w);
  HIPCHK(hipMemsetAsync(d_out, 0, ((size_t)batch << n_keep) * sizeof(float), stream));
  const uint64_t D = (uint64_t)1 << n_qubits;
  if (n_keep <= 12) {
    // (<= 512 workgroups per state: >= 2^n / 512 terms are summed in LDS per global atomic)
    hipLaunchKernelGGL(k_marginal_lds, dim3(grid_for(D, 256, 512), batch), dim3(256),
                       sizeof(float) << n_keep, stream, (const float2 *)d_states, d_out, n_qubits, kb);
  } else {
    hipLaunchKernelGGL(k_marginal, dim3(grid_for(D, 256, 4096), batch), dim3(256), 0, stream,
                       (const float2 *)d_states, d_out, n_qubits, kb);
  }
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

static int overlap_blocks(int n) {
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  uint64_t b = (chunks + 256 * 4 - 1) / (256 * 4);
  if (b < 1) b = 1;
  if (b > 1024) b = 1024;
  return (int)b;
}

size_t qmle_pair_fidelity_workspace_bytes(int n_qubits, int n_pairs) {
  if (n_qubits < 1 || n_pairs < 1) return 0;
  return (size_t)n_pairs * overlap_blocks(n_qubits) * sizeof(float2) + 256;
}

int qmle_pair_fidelity(const void *d_states, int n_qubits, int n_pairs, float *d_out,
                       void *d_workspace, size_t workspace_bytes, qmle_stream stream_) {
  if (!d_states || !d_out || !d_workspace || n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS ||
      n_pairs < 1)
    return QMLE_ERR_INVALID_ARG;
  const int nb = overlap_blocks(n_qubits);
  if (workspace_bytes < (size_t)n_pairs * nb * sizeof(float2)) return QMLE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const uint64_t chunks = (uint64_t)1 << (n_qubits - 1);
  for (int p0 = 0; p0 < n_pairs; p0 += 65535) {
    const int pc = n_pairs - p0 < 65535 ? n_pairs - p0 : 65535;
    // pairs (i, i + n_pairs): shift both halves by p0
    hipLaunchKernelGGL(k_overlap_partial, dim3(nb, pc), dim3(256), 0, stream,
                       (const float4 *)d_states + (size_t)p0 * chunks, n_qubits, n_pairs,
                       (float2 *)d_workspace + (size_t)p0 * nb);
  }
  hipLaunchKernelGGL(k_overlap_final, dim3(n_pairs), dim3(nb >= 256 ? 256 : 64), 0, stream,
                     (const float2 *)d_workspace, nb, n_pairs, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_density_probs(const void *d_rho, int n_qubits, int batch, float *d_out,
                       qmle_stream stream) {
  if (!d_rho || !d_out || n_qubits < 1 || 2 * n_qubits > QMLE_MAX_QUBITS || batch < 1 ||
      batch > 65535)
    return QMLE_ERR_INVALID_ARG;
  const uint64_t D = (uint64_t)1 << n_qubits;
  hipLaunchKernelGGL(k_density_probs, dim3(grid_for(D, 256), batch), dim3(256), 0,
                     (hipStream_t)stream, (const float2 *)d_rho, n_qubits, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_density_expval_z(const void *d_rho, int n_qubits, int batch, const int32_t *obs_wires,
                          int n_obs, float *d_out, qmle_stream stream) {
  if (!d_rho || !d_out || !obs_wires || n_qubits < 1 || 2 * n_qubits > QMLE_MAX_QUBITS ||
      batch < 1 || batch > 65535 || n_obs < 1 || n_obs > QMLE_MAX_QUBITS)
    return QMLE_ERR_INVALID_ARG;
  ObsBits ob;
  for (int k = 0; k < n_obs; ++k) {
    if (obs_wires[k] < 0 || obs_wires[k] >= n_qubits) return QMLE_ERR_WIRE_RANGE;
    ob.bits[k] = (int8_t)(n_qubits - 1 - obs_wires[k]);
  }
  hipLaunchKernelGGL(k_density_expval, dim3(batch, n_obs), dim3(256), 0, (hipStream_t)stream,
                     (const float2 *)d_rho, n_qubits, ob, n_obs, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

size_t qmle_overlap_workspace_bytes(int n_qubits, int count) {
  if (n_qubits < 1 || count < 1) return 0;
  return (size_t)count * overlap_blocks(n_qubits) * sizeof(float2) + 256;
}

int qmle_overlap(const void *d_a, const void *d_b, int n_qubits, int count, void *d_out,
                 void *d_workspace, size_t workspace_bytes, qmle_stream stream_) {
  if (!d_a || !d_b || !d_out || !d_workspace || n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS ||
      count < 1)
    return QMLE_ERR_INVALID_ARG;
  const int nb = overlap_blocks(n_qubits);
  if (workspace_bytes < (size_t)count * nb * sizeof(float2)) return QMLE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const uint64_t chunks = (uint64_t)1 << (n_qubits - 1);
  for (int p0 = 0; p0 < count; p0 += 65535) {
    const int pc = count - p0 < 65535 ? count - p0 : 65535;
    hipLaunchKernelGGL(k_overlap2_partial, dim3(nb, pc), dim3(256), 0, stream,
                       (const float4 *)d_a + (size_t)p0 * chunks,
                       (const float4 *)d_b + (size_t)p0 * chunks, n_qubits,
                       (float2 *)d_workspace + (size_t)p0 * nb);
  }
  hipLaunchKernelGGL(k_overlap2_final, dim3(count), dim3(nb >= 256 ? 256 : 64), 0, stream,
                     (const float2 *)d_workspace, nb, count, (float2 *)d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

size_t qmle_expval_parity_workspace_bytes(int n_qubits, int batch) {
  if (n_qubits < 1 || batch < 1) return 0;
  return (size_t)batch * overlap_blocks(n_qubits) * 8 * sizeof(float) + 256;
}

int qmle_expval_parity(const void *d_states, int n_qubits, int batch, const uint32_t *wire_masks,
                       int n_obs, float *d_out, void *d_workspace, size_t workspace_bytes,
                       qmle_stream stream_) {
  if (!d_states || !d_out || !d_workspace || !wire_masks || n_qubits < 1 ||
      n_qubits > QMLE_MAX_QUBITS || batch < 1 || batch > 65535 || n_obs < 1)
    return QMLE_ERR_INVALID_ARG;
  const int nb = overlap_blocks(n_qubits);
  if (workspace_bytes < (size_t)batch * nb * 8 * sizeof(float)) return QMLE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  for (int o0 = 0; o0 < n_obs; o0 += 8) {
    ParityMasks pm;
    pm.count = n_obs - o0 < 8 ? n_obs - o0 : 8;
    for (int k = 0; k < 8; ++k) {
      uint32_t bits = 0;
      if (k < pm.count) {
        const uint32_t wm = wire_masks[o0 + k];  // bit w set <=> wire w in the parity
        if (n_qubits < 32 && (wm >> n_qubits)) return QMLE_ERR_WIRE_RANGE;
        for (int w = 0; w < n_qubits; ++w)
          if (wm & (1u << w)) bits |= 1u << (n_qubits - 1 - w);
      }
      pm.m[k] = bits;
    }
    hipLaunchKernelGGL(k_parity_partial, dim3(nb, batch), dim3(256), 0, stream,
                       (const float4 *)d_states, n_qubits, pm, (float *)d_workspace);
    hipLaunchKernelGGL(k_parity_final, dim3(batch), dim3(nb >= 256 ? 256 : 64), 0, stream,
                       (const float *)d_workspace, nb, pm.count, n_obs, o0, d_out);
  }
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

// Plan of the reads for an n-qubit state (host only).  Positions >= 12 are cut into 4-bit chunks
// [12,16), [16,20), ... with the last one aligned to the top [n-4, n); a later read takes two
// chunks, paired outermost with innermost -- at n = 28: {12-15, 24-27} and {16-19, 20-23}, the
// pairs that stream fastest (see the note above k_mw_read).  An odd chunk is paired with a
// lower, already reported one.  A chunk may overlap its neighbour (n not a multiple of 4): every
// position takes its sums from the first read that reports it.
static MwPlan mw_plan(int n, int batch) {
  MwPlan pl;
  std::memset(&pl, 0, sizeof(pl));
  for (int p = 0; p < n; ++p) pl.src_read[p] = -1;
  for (int p = 0; p < kMwT && p < n; ++p) { pl.src_read[p] = 0; pl.src_col[p] = p; }
  const uint32_t tiles = 1u << (n - kMwT);
  // tiles per workgroup: the first read keeps ~40 sums per work item, a long walk amortises its
  // reduction (2^4); the later reads stream best at 2^2 (tools/mw_tune.hip); never fewer than
  // ~2048 workgroups per launch
  auto pick_q = [&](int want) {
    int q = 0;
    while (q < want && (((uint64_t)batch * tiles) >> (q + 1)) >= 2048) ++q;
    return q;
  };
  static const int q_env = std::getenv("QMLE_MW_Q") ? atoi(std::getenv("QMLE_MW_Q")) : -1;
  pl.q_first = pick_q(4);
  if (q_env >= 0 && q_env <= 4 && (tiles >> q_env) >= 1) pl.q_first = q_env;
  pl.rows_first = tiles >> pl.q_first;
  int chunks[8], nc = 0;
  for (int c = kMwT; c < n; c += 4) chunks[nc++] = c + 4 <= n ? c : n - 4;
  int i = 0, j = nc - 1;
  while (i <= j) {
    int a = chunks[i], b = i < j ? chunks[j] : -1;
    if (b < 0) {  // odd one out: pair it with a lower, already reported chunk
      b = a;
      a = b >= 16 ? 8 : b - 4;
    }
    if (a > b) std::swap(a, b);
    if (b < a + 4) a = b - 4;  // overlapping chunks (n not a multiple of 4): shift the lower one down
    const int r = pl.n_later++;
    pl.lo[r] = a;
    pl.lo2[r] = b;
    pl.q[r] = pick_q(2);
    if (q_env >= 0 && q_env <= 4 && (tiles >> q_env) >= 1) pl.q[r] = q_env < 2 ? q_env : 2;
    pl.rows_later[r] = tiles >> pl.q[r];
    for (int k = 0; k < 8; ++k) {
      const int p = k < 4 ? a + k : b + k - 4;
      if (p < n && pl.src_read[p] < 0) { pl.src_read[p] = r + 1; pl.src_col[p] = k; }
    }
    ++i;
    --j;
  }
  return pl;
}

int qmle_meyer_wallach_reads(int n_qubits) {
  if (n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS) return 0;
  // below the tile size: one (cache-resident) sweep per wire
  return n_qubits >= kMwT ? 1 + mw_plan(n_qubits, 1).n_later : n_qubits;
}

size_t qmle_meyer_wallach_workspace_bytes(int n_qubits, int batch) {
  if (n_qubits < 1 || batch < 1) return 0;
  if (n_qubits >= kMwT) {
    const MwPlan pl = mw_plan(n_qubits, batch);
    size_t fl = (size_t)pl.rows_first * kMwRowFirst;
    for (int r = 0; r < pl.n_later; ++r) fl += (size_t)pl.rows_later[r] * kMwRowLater;
    return ((size_t)batch * fl + (size_t)batch * QMLE_MAX_QUBITS) * sizeof(float) + 512;
  }
  return (size_t)batch * n_qubits * overlap_blocks(n_qubits) * sizeof(float4) + 256;
}

int qmle_meyer_wallach(const void *d_states, int n_qubits, int batch, float *d_out,
                       float *d_purities, void *d_workspace, size_t workspace_bytes,
                       qmle_stream stream_) {
  if (!d_states || !d_out || !d_workspace || n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS ||
      batch < 1 || batch > 65535)
    return QMLE_ERR_INVALID_ARG;
  if (workspace_bytes + 256 < qmle_meyer_wallach_workspace_bytes(n_qubits, batch))
    return QMLE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int n = n_qubits;
  if (n >= kMwT) {  // LDS-staged tiles: 1 + ceil((n - 12) / 8) reads of the state
    if (first_use_on_device(5)) {
      QMLE_LDS_BASE_CHECK(k_mw_read_first<true>);
      QMLE_LDS_BASE_CHECK(k_mw_read_first<false>);
      QMLE_LDS_BASE_CHECK(k_mw_read_later<true>);
      QMLE_LDS_BASE_CHECK(k_mw_read_later<false>);
    }
    const MwPlan pl = mw_plan(n, batch);
    for (int p = 0; p < n; ++p)
      if (pl.src_read[p] < 0) return QMLE_ERR_INTERNAL;
    const uint32_t tiles = 1u << (n - kMwT);
    // >= 1 GiB per launch: stream past the caches
    const bool nt = ((uint64_t)batch << (n + 3)) >= (1ull << 30);
    const size_t lds = (size_t)8 << kMwT;
    MwPurityArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    float *ws = (float *)d_workspace;
    MwReadArgs a;
    a.states = (const float2 *)d_states;
    a.n = n;
    a.rows = ws;
    a.lo = 4;
    a.lo2 = 8;
    a.q = pl.q_first;
    pa.first = ws;
    pa.rows_first = pl.rows_first;
    pa.q_first = pl.q_first;
    pa.n = n;
    ws += (size_t)batch * pl.rows_first * kMwRowFirst;
    // The reads are independent of each other.  The first read is the arithmetic-heavy one (192
    // packed fmas + the population butterfly per 16 amplitudes) and is the one that suffers when
    // it starts on a chip that has just idled through the tiny reduction kernels of a previous
    // call (0.33 ms warm, 0.42 - 0.47 ms cold at n = 28); the later reads are bound by HBM
    // alone.  So it runs LAST (QMLE_MW_FIRST_FIRST=1 for the A/B).
    static const bool first_first = std::getenv("QMLE_MW_FIRST_FIRST") != nullptr;
    auto launch_first = [&]() {
      const dim3 grid(tiles >> a.q, batch);
      if (nt) hipLaunchKernelGGL(k_mw_read_first<true>, grid, dim3(kMwThreads), lds, stream, a);
      else hipLaunchKernelGGL(k_mw_read_first<false>, grid, dim3(kMwThreads), lds, stream, a);
    };
    const MwReadArgs a_first = a;
    if (first_first || pl.n_later == 0) launch_first();
    for (int r = 0; r < pl.n_later; ++r) {
      a.rows = ws;
      a.lo = pl.lo[r];
      a.lo2 = pl.lo2[r];
      a.q = pl.q[r];
      pa.later[r] = ws;
      pa.rows_later[r] = pl.rows_later[r];
      ws += (size_t)batch * pl.rows_later[r] * kMwRowLater;
      const dim3 grid(tiles >> a.q, batch);
      if (nt) hipLaunchKernelGGL(k_mw_read_later<true>, grid, dim3(kMwThreads), lds, stream, a);
      else hipLaunchKernelGGL(k_mw_read_later<false>, grid, dim3(kMwThreads), lds, stream, a);
    }
    if (!first_first && pl.n_later > 0) {
      a = a_first;
      launch_first();
    }
    for (int p = 0; p < n; ++p) { pa.src_read[p] = (int8_t)pl.src_read[p]; pa.src_col[p] = (int8_t)pl.src_col[p]; }
    float *d_pur = ws;
    hipLaunchKernelGGL(k_mw_purity, dim3(batch, n), dim3(pl.rows_first >= 1024 ? 1024 : pl.rows_first >= 256 ? 256 : 64),
                       0, stream, pa, d_pur);
    hipLaunchKernelGGL(k_mw_tile_q, dim3((batch + 63) / 64), dim3(64), 0, stream,
                       (const float *)d_pur, n, batch, d_out, d_purities);
    HIPCHK(hipGetLastError());
    return QMLE_OK;
  }
  const int nb = overlap_blocks(n_qubits);
  for (int p = 0; p < n_qubits; ++p)
    hipLaunchKernelGGL(k_cross_partial, dim3(nb, batch), dim3(256), 0, stream,
                       (const float4 *)d_states, n_qubits, p, (float4 *)d_workspace, nb);
  hipLaunchKernelGGL(k_mw_final, dim3(batch), dim3(nb >= 256 ? 256 : 64), 0, stream,
                     (const float4 *)d_workspace, n_qubits, nb, batch, d_out, d_purities);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_philox_uniform_f32_device(const uint64_t key[2], uint64_t n, double low, double high, float *d_out,
                                   qmle_stream stream_) {
  if (!key || (!d_out && n > 0) || n > (1ull << 40)) return QMLE_ERR_INVALID_ARG;
  if (n == 0) return QMLE_OK;
  hipLaunchKernelGGL(k_philox_uniform, dim3(grid_for((n + 3) / 4, 256, 1u << 16)), dim3(256), 0, (hipStream_t)stream_,
                     key[0], key[1], n, low, high - low, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_histogram(const float *d_values, int64_t count, int n_bins, float lo, float hi,
                   int32_t *d_counts, qmle_stream stream_) {
  if (!d_values || !d_counts || count < 0 || n_bins < 1 || !(hi > lo)) return QMLE_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  HIPCHK(hipMemsetAsync(d_counts, 0, (size_t)n_bins * sizeof(int32_t), stream));
  if (count > 0)
    hipLaunchKernelGGL(k_histogram, dim3(grid_for((uint64_t)count, 256, 1024)), dim3(256), 0,
                       stream, d_values, count, n_bins, lo, hi, d_counts);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

size_t qmle_sample_workspace_bytes(int n_qubits, int batch) {
  if (n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS || batch < 1) return 0;
  return ((size_t)batch << n_qubits) * sizeof(double);
}

int qmle_sample_counts(const float *d_probs, int n_qubits, int batch, int shots, uint64_t seed,
                       uint64_t row_offset, int32_t *d_counts, float *d_est_probs,
                       void *d_workspace, size_t workspace_bytes, qmle_stream stream_) {
  if (!d_probs || !d_counts || !d_workspace || n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS ||
      batch < 1 || batch > 65535 || shots < 1)
    return QMLE_ERR_INVALID_ARG;
  if (workspace_bytes < qmle_sample_workspace_bytes(n_qubits, batch))
    return QMLE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const uint64_t D = (uint64_t)1 << n_qubits;
  double *cdf = (double *)d_workspace;
  HIPCHK(hipMemsetAsync(d_counts, 0, (size_t)batch * D * sizeof(int32_t), stream));
  hipLaunchKernelGGL(k_cdf, dim3(batch), dim3(256), 0, stream, d_probs, D, cdf);
  const unsigned gx = (unsigned)(((int64_t)shots + kShotsPerBlock - 1) / kShotsPerBlock);
  if (D <= (uint64_t)kLdsHistMax)
    hipLaunchKernelGGL(k_sample<true>, dim3(gx, batch), dim3(256), 0, stream, cdf, D, shots,
                       seed, row_offset, d_counts);
  else
    hipLaunchKernelGGL(k_sample<false>, dim3(gx, batch), dim3(256), 0, stream, cdf, D, shots,
                       seed, row_offset, d_counts);
  if (d_est_probs)
    hipLaunchKernelGGL(k_counts_to_probs, dim3(grid_for((uint64_t)batch * D, 256, 4096)),
                       dim3(256), 0, stream, d_counts, (uint64_t)batch * D, 1.0f / (float)shots,
                       d_est_probs);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_probs_diag_expval(const float *d_probs, int n_qubits, int batch,
                           const int32_t *obs_wires, const int32_t *obs_n_wires,
                           const int32_t *obs_diag_off, const float *d_diag, int n_obs,
                           float *d_out, void *d_workspace, size_t workspace_bytes,
                           qmle_stream stream_) {
  if (!d_probs || !d_out || !obs_wires || !obs_n_wires || !obs_diag_off || !d_workspace ||
      n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS || batch < 1 || batch > 65535 || n_obs < 1 ||
      n_obs > 65535)
    return QMLE_ERR_INVALID_ARG;
  if (workspace_bytes < (size_t)n_obs * sizeof(DiagObs)) return QMLE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  std::vector<DiagObs> host(n_obs);
  int w0 = 0;
  for (int k = 0; k < n_obs; ++k) {
    const int nw = obs_n_wires[k];
    if (nw < 1 || nw > n_qubits) return QMLE_ERR_INVALID_ARG;
    uint32_t seen = 0;
    for (int j = 0; j < nw; ++j) {
      const int w = obs_wires[w0 + j];
      if (w < 0 || w >= n_qubits) return QMLE_ERR_WIRE_RANGE;
      if (seen & (1u << w)) return QMLE_ERR_DUPLICATE_WIRES;
      seen |= 1u << w;
      host[k].bits[j] = (int8_t)(n_qubits - 1 - w);
    }
    host[k].n_wires = nw;
    host[k].diag_off = obs_diag_off[k];
    if (host[k].diag_off >= 0 && !d_diag) return QMLE_ERR_INVALID_ARG;
    w0 += nw;
  }
  HIPCHK(hipMemcpyAsync(d_workspace, host.data(), (size_t)n_obs * sizeof(DiagObs),
                        hipMemcpyHostToDevice, stream));
  HIPCHK(hipStreamSynchronize(stream));  // `host` dies with this frame
  hipLaunchKernelGGL(k_probs_diag_expval, dim3(batch, n_obs), dim3(256), 0, stream, d_probs,
                     (uint64_t)1 << n_qubits, (const DiagObs *)d_workspace, d_diag, n_obs, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

size_t qmle_probs_diag_expval_workspace_bytes(int n_obs) {
  return n_obs < 1 ? 0 : (size_t)n_obs * sizeof(DiagObs);
}

}  // extern "C"

// ---- complex128 engine: host side ------------------------------------------------------------
static int ensure_f64(qmle_plan *p) {
  if (p->f64_blob) return p->f64_device == current_device() ? QMLE_OK : QMLE_ERR_UNSUPPORTED;
  const size_t b_low = align_up(p->lowered.size() * sizeof(LoweredOp) + 16, 256);
  const size_t nc = p->n_user_consts;
  const size_t b_c = align_up(nc * sizeof(double) + 16, 256);
  char *blob = nullptr;
  HIPCHK(hipMalloc((void **)&blob, b_low + b_c));
  p->f64_blob = blob;
  p->f64_device = current_device();
  if (!p->lowered.empty())
    HIPCHK(hipMemcpy(blob, p->lowered.data(), p->lowered.size() * sizeof(LoweredOp), hipMemcpyHostToDevice));
  if (nc) {
    std::vector<double> c64(nc);
    for (size_t i = 0; i < nc; ++i) c64[i] = i < p->consts64.size() ? p->consts64[i] : (double)p->consts[i];
    HIPCHK(hipMemcpy(blob + b_low, c64.data(), nc * sizeof(double), hipMemcpyHostToDevice));
  }
  return QMLE_OK;
}

static size_t f64_states_in_flight(const qmle_plan *p, int batch) {
  const size_t sb = (size_t)16 << p->n;
  size_t s = ((size_t)4 << 30) / sb;  // 4 GiB of states per round of launches
  if (s < 1) s = 1;
  if (s > (size_t)batch) s = (size_t)batch;
  if (s > 65535) s = 65535;
  return s;
}

extern "C" {

int qmle_plan_set_consts_f64(qmle_plan *plan, const double *consts, int n_consts) {
  if (!plan || n_consts < 0 || (n_consts > 0 && !consts) || (size_t)n_consts != plan->n_user_consts)
    return QMLE_ERR_INVALID_ARG;
  if (plan->f64_blob) return QMLE_ERR_UNSUPPORTED;  // before the first complex128 run
  plan->consts64.assign(consts, consts + n_consts);
  return QMLE_OK;
}

size_t qmle_workspace_bytes_f64(const qmle_plan *plan, int batch, int meas_type) {
  if (!plan || batch < 1) return 0;
  size_t total = align_up((size_t)batch * (plan->mat_floats ? plan->mat_floats : 1) * sizeof(double), 256) + 512;
  const bool lds = plan->n <= 13;
  if (!lds || meas_type == QMLE_MEAS_DENSITY)
    total += (lds ? (size_t)batch : f64_states_in_flight(plan, batch)) * align_up((size_t)16 << plan->n, 256);
  return total;
}

// complex128 counterpart of qmle_apply_inplace: the plan's operators, one launch each, on resident
// states (no initialisation, no measurement) -- what the doubled-register density path needs
// between two Kraus channels (simulation.py:107-128, operations.py:485-512, 1551-1578)
size_t qmle_apply_inplace_f64_workspace_bytes(const qmle_plan *plan, int batch) {  // the matrix rows
  if (!plan || batch < 1) return 0;
  return align_up((size_t)batch * (plan->mat_floats ? plan->mat_floats : 1) * sizeof(double), 256) + 512;
}
int qmle_apply_inplace_f64(qmle_plan *plan, const double *d_angles, int batch, void *d_states,
                           void *d_workspace, size_t workspace_bytes, qmle_stream stream_) {
  if (!plan || batch < 1 || batch > 65535 || !d_states || !d_workspace) return QMLE_ERR_INVALID_ARG;
  if (plan->n_slots > 0 && !d_angles) return QMLE_ERR_INVALID_ARG;
  if (workspace_bytes < qmle_apply_inplace_f64_workspace_bytes(plan, batch)) return QMLE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  int rc = ensure_device_plan(plan);
  if (rc != QMLE_OK) return rc;
  rc = ensure_f64(plan);
  if (rc != QMLE_OK) return rc;
  const double *d_c64 = (const double *)((char *)plan->f64_blob + align_up(plan->lowered.size() * sizeof(LoweredOp) + 16, 256));
  char *ws = (char *)d_workspace;
  ws += (size_t)(256 - ((uintptr_t)ws & 255)) & 255;
  double *d_mats = (double *)ws;
  if (!plan->groups.empty()) {
    const int ng = (int)plan->groups.size();
    hipLaunchKernelGGL(k_build_matrices_f64, dim3((ng + 63) / 64, batch), dim3(64), 0, stream, plan->dev.d_build,
                       plan->dev.d_groups, ng, d_angles, plan->n_slots, d_c64, d_mats, plan->mat_floats);
  }
  const int n = plan->n;
  const unsigned gx = grid_for(((size_t)1 << n) / 2, 256, 1u << 16);
  for (size_t k = 0; k < plan->lowered.size(); ++k)
    hipLaunchKernelGGL(k64_op, dim3(gx ? gx : 1, batch), dim3(256), 0, stream, (double2 *)d_states, n, plan->lowered[k],
                       d_mats, plan->mat_floats, d_c64, d_angles, plan->n_slots);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_run_batch_f64(qmle_plan *plan, const double *d_angles, int batch, int meas_type,
                       const uint32_t *wire_masks, int n_obs, void *d_out, void *d_workspace,
                       size_t workspace_bytes, qmle_stream stream_) {
  if (!plan || batch < 1 || !d_out || !d_workspace) return QMLE_ERR_INVALID_ARG;
  if (meas_type < QMLE_MEAS_STATE || meas_type > QMLE_MEAS_DENSITY) return QMLE_ERR_MEAS_TYPE;
  if (plan->n_slots > 0 && !d_angles) return QMLE_ERR_INVALID_ARG;
  if (workspace_bytes < qmle_workspace_bytes_f64(plan, batch, meas_type)) return QMLE_ERR_WORKSPACE;
  const int n = plan->n;
  F64Obs obs;
  std::memset(&obs, 0, sizeof(obs));
  if (meas_type == QMLE_MEAS_EXPVAL_Z) {
    if (n_obs < 1 || n_obs > QMLE_MAX_QUBITS || !wire_masks) return QMLE_ERR_INVALID_ARG;
    for (int k = 0; k < n_obs; ++k) {
      const uint32_t in = wire_masks[k];
      if (in == 0 || (n < 32 && (in >> n))) return QMLE_ERR_WIRE_RANGE;
      for (int w = 0; w < n; ++w)
        if (in & (1u << w)) obs.mask[k] |= 1u << (n - 1 - w);
    }
  }
  if (meas_type == QMLE_MEAS_DENSITY && n > 12) return QMLE_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  int rc = ensure_device_plan(plan);
  if (rc != QMLE_OK) return rc;
  rc = ensure_f64(plan);
  if (rc != QMLE_OK) return rc;
  const LoweredOp *d_low = (const LoweredOp *)plan->f64_blob;
  const double *d_c64 = (const double *)((char *)plan->f64_blob + align_up(plan->lowered.size() * sizeof(LoweredOp) + 16, 256));
  char *ws = (char *)d_workspace;
  const size_t mis = (size_t)(256 - ((uintptr_t)ws & 255)) & 255;
  ws += mis;
  double *d_mats = (double *)ws;
  ws += align_up((size_t)batch * (plan->mat_floats ? plan->mat_floats : 1) * sizeof(double), 256);
  if (!plan->groups.empty()) {
    const int ng = (int)plan->groups.size();
    for (int b0 = 0; b0 < batch; b0 += 65535) {
      const int bc = batch - b0 < 65535 ? batch - b0 : 65535;
      hipLaunchKernelGGL(k_build_matrices_f64, dim3((ng + 63) / 64, bc), dim3(64), 0, stream, plan->dev.d_build,
                         plan->dev.d_groups, ng, d_angles + (size_t)b0 * plan->n_slots, plan->n_slots, d_c64,
                         d_mats + (size_t)b0 * plan->mat_floats, plan->mat_floats);
    }
  }
  const size_t D = (size_t)1 << n;
  const int n_ops = (int)plan->lowered.size();
  double2 *d_states = (double2 *)ws;
  if (n <= 13) {
    if (first_use_on_device(6))
      HIPCHK(hipFuncSetAttribute((const void *)k64_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    void *target = meas_type == QMLE_MEAS_DENSITY ? (void *)d_states : d_out;
    hipLaunchKernelGGL(k64_lds, dim3(batch), dim3(256), D * sizeof(double2), stream, d_low, n_ops, n, d_mats,
                       plan->mat_floats, d_c64, d_angles, plan->n_slots, meas_type, obs, n_obs, target);
    if (meas_type == QMLE_MEAS_DENSITY)
      for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int bc = batch - b0 < 65535 ? batch - b0 : 65535;
        hipLaunchKernelGGL(k64_density, dim3(grid_for(D * D, 256, 1u << 16), bc), dim3(256), 0, stream,
                           d_states + (size_t)b0 * D, n, (double2 *)d_out + (size_t)b0 * D * D);
      }
    HIPCHK(hipGetLastError());
    return QMLE_OK;
  }
  const int in_flight = (int)f64_states_in_flight(plan, batch);
  for (int b0 = 0; b0 < batch; b0 += in_flight) {
    const int bc = batch - b0 < in_flight ? batch - b0 : in_flight;
    double2 *stc = meas_type == QMLE_MEAS_STATE ? (double2 *)d_out + (size_t)b0 * D : d_states;
    const unsigned gx = grid_for(D / 2, 256, 1u << 16);
    hipLaunchKernelGGL(k64_init, dim3(gx, bc), dim3(256), 0, stream, stc, n);
    for (int k = 0; k < n_ops; ++k)
      hipLaunchKernelGGL(k64_op, dim3(gx, bc), dim3(256), 0, stream, stc, n, plan->lowered[k],
                         d_mats + (size_t)b0 * plan->mat_floats, plan->mat_floats, d_c64,
                         d_angles ? d_angles + (size_t)b0 * plan->n_slots : nullptr, plan->n_slots);
    if (meas_type == QMLE_MEAS_PROBS)
      hipLaunchKernelGGL(k64_probs, dim3(grid_for((uint64_t)bc * D, 256, 1u << 20)), dim3(256), 0, stream, stc,
                         (double *)d_out + (size_t)b0 * D, (uint64_t)bc * D);
    else if (meas_type == QMLE_MEAS_EXPVAL_Z)
      hipLaunchKernelGGL(k64_expval, dim3(bc, n_obs), dim3(256), 0, stream, stc, n, obs, n_obs,
                         (double *)d_out + (size_t)b0 * n_obs);
    HIPCHK(hipGetLastError());
  }
  return QMLE_OK;
}

}  // extern "C"
