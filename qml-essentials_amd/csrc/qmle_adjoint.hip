// libqmle_sv, adjoint differentiation: the whole sweep in LDS (n <= 13), fused tile passes over
// psi and lambda (n >= 14), per-gate streaming overlaps, and qmle_adjoint_gradient.
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


}  // namespace

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

extern "C" {

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
  // (the sweep applies stages to LIVE states: a schedule compiled for runs from |0..0> only is refused)
  if ((fwd->flags | rev->flags) & QMLE_PLAN_INTERNAL_ZERO_RUN) return QMLE_ERR_UNSUPPORTED;
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
    rc = launch_build_matrices(fwd, d_angles_fwd, fmats, batch, stream);
    if (rc == QMLE_OK) rc = launch_build_matrices(rev, d_angles_rev, rmats, batch, stream);
    if (rc != QMLE_OK) return rc;
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
    if (FirstUse once{3}; once.first) {
      HIPCHK(hipFuncSetAttribute((const void *)k_adjoint_lds<false>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void *)k_adjoint_lds<true>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      once.done();
    }
    bool has_dense4 = false;
    for (int g = fst.grp_begin; g < fst.grp_end; ++g) has_dense4 |= fwd->op_groups[g].kind == GK_DENSE4 || fwd->op_groups[g].kind == GK_REG4X;
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
  rc = launch_build_matrices(rev, ang2, mats, 2 * batch, stream);
  if (rc != QMLE_OK) return rc;
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
    if (FirstUse once{4}; once.first) {
      HIPCHK(hipFuncSetAttribute((const void *)k_tile_adj,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      once.done();
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

}  // extern "C"
