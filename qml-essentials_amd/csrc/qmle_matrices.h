// Per-sample gate matrices from the angle table (templates shared by the complex64 engine and the
// complex128 one): operations.py:1002-1045, 1053-1100, 1171-1243, 1255-1351, 1357-1487 restated.
#pragma once
#include "qmle_dev.h"

namespace {

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
// ANG: anything indexable by slot -- a row of the angle table (`const float *` / `const double *`) or an
// AngleMapRow that forms the angle from the leaves on the fly (same arithmetic as k_build_angles)
template <class ANG, class CT>
__device__ __forceinline__ M2 source_2x2(const BuildOp &b, ANG ang, const CT *__restrict__ consts) {
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

template <class ANG, class CT>
__device__ void source_matrix(const BuildOp &b, ANG ang,
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

// Angle table computed on the fly: table[b][s] = c[s] + sum_t coef[t] * leaf_{arg[t]}[row_k(b)][idx[t]],
// row_k(b) = ((b + b_offset) / div_k) % mod_k, fp64 sum, reduced into (-period/2, period/2], rounded to
// float32 -- bit for bit what k_build_angles stores (qmle_engine.hip); qmle_run_batch_map uses it to skip
// that kernel and the table's round trip when nothing but the matrix builder reads angles.
struct AngleLeaves {
  const float *ptr[8];
  long long stride[8];  // floats per row
  int div[8], mod[8];
};
struct AngleMapSrc {
  AngleLeaves lv;
  const int *ptr, *arg, *idx;
  const float *coef, *cst;
  const double *period;
  long long b_offset;
  int small;  // batch + offset < 2^32: row arithmetic in 32 bits
};
struct AngleMapRow {
  const AngleMapSrc *m;
  unsigned long long gb;
  __device__ __forceinline__ float operator[](int s) const {
    double acc = (double)m->cst[s];
    for (int t = m->ptr[s]; t < m->ptr[s + 1]; ++t) {
      const int k = m->arg[t];
      const long long row = m->small ? (long long)(((uint32_t)gb / (uint32_t)m->lv.div[k]) % (uint32_t)m->lv.mod[k])
                                     : (long long)((gb / (unsigned long long)m->lv.div[k]) % (unsigned long long)m->lv.mod[k]);
      acc = fma((double)m->coef[t], (double)m->lv.ptr[k][row * m->lv.stride[k] + m->idx[t]], acc);
    }
    const double per = m->period ? m->period[s] : 0.0;
    if (per > 0.0 && (acc > per || acc < -per)) acc -= per * rint(acc / per);
    return (float)acc;
  }
};
template <class T>
__device__ __forceinline__ const T *angle_row(const T *table, int b, int n_slots) { return table + (size_t)b * n_slots; }
__device__ __forceinline__ AngleMapRow angle_row(const AngleMapSrc &m, int b, int) {
  return AngleMapRow{&m, (unsigned long long)((long long)b + m.b_offset)};
}

// AT / CT / OT: angle table, constant blob and matrix row types -- float / float / float for the
// complex64 engine, double throughout for the complex128 one (qmle_run_batch_f64)
// GMAJOR: whole waves per group (launch side: batch >= 64, blocks of one wave) -- the group index is then
// provably wave-uniform and the descriptors come through the scalar cache
// ASRC: `const AT *` (the angle table) or `const AngleMapSrc &`
template <class ASRC, class CT, class OT, bool GMAJOR = false>
__device__ __forceinline__ void build_matrices_body(const BuildOp *__restrict__ build,
                                                    const BuildGroup *__restrict__ groups, int n_groups,
                                                    ASRC angles, int n_slots,
                                                    const CT *__restrict__ consts, OT *__restrict__ mats,
                                                    uint32_t mat_floats, int batch) {
  // one work item per (sample, group).  batch >= 64 (blocks of ONE wave): a wave takes 64 consecutive
  // samples of the SAME group -- every lane then walks the same source gates through the same branches
  // of the opcode switch.  With the samples-then-groups flattening used below that, neighbouring lanes
  // held different groups (RX next to RY next to CX ...) and a wave executed every branch any of its lanes
  // needed, one after the other: 20 us for the 4096 x 130 groups of the Fourier grid, untouched by cheaper
  // trigonometry, fewer registers, prefetched descriptors or LDS staging.
  int b, g;
  if constexpr (GMAJOR) {
    const uint32_t per = ((uint32_t)batch + 63u) >> 6;  // waves per group
    g = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / per));
    b = (int)((blockIdx.x - (uint32_t)g * per) * 64u + threadIdx.x);
    if (g >= n_groups || b >= batch) return;
  } else {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long b_ll = idx / n_groups;
    if (b_ll >= batch) return;
    b = (int)b_ll;
    g = (int)(idx - b_ll * n_groups);
  }
  const BuildGroup grp = groups[g];
  const auto ang = angle_row(angles, b, n_slots);
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
  // (BuildOp::pad = 1 / 2: a 2x2 source on the first / second wire of the pair, U (x) I / I (x) U -- row = 2 * bit[t0] + bit[t1])
  auto source4 = [&](const BuildOp &bo, cd *X) {
    if (dim != 4 || bo.pad == 0) {
      source_matrix(bo, ang, consts, X, dim);
      return;
    }
    const M2 U = source_2x2(bo, ang, consts);
    for (int i = 0; i < 16; ++i) X[i] = {0.0, 0.0};
    if (bo.pad == 1) {
      X[0] = U.a; X[2] = U.b; X[5] = U.a; X[7] = U.b; X[8] = U.c; X[10] = U.d; X[13] = U.c; X[15] = U.d;
    } else {
      X[0] = U.a; X[1] = U.b; X[4] = U.c; X[5] = U.d; X[10] = U.a; X[11] = U.b; X[14] = U.c; X[15] = U.d;
    }
  };
  source4(build[grp.begin], M);
  for (uint32_t k = grp.begin + 1; k < grp.end; ++k) {
    source4(build[k], S);
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

}  // namespace
