// Device side of an LDS tile pass, shared by the tile kernels (qmle_tile.hip) and the adjoint
// sweep (qmle_adjoint.hip): tile arguments, the XOR-swizzled LDS layout, the register-tile gate
// appliers, the generic gate sweeps and the store / measure epilogues.
#pragma once
#include <cstring>

#include "qmle_dev.h"
#include "qmle_host.h"

namespace {

// ---------------------------------------------------------------------------
// LDS tile kernel
// ---------------------------------------------------------------------------
// (enum TileMeas: qmle_host.h -- the engine picks the epilogue)

// (sw(), the XOR swizzle of the LDS tile layout, lives in qmle_dev.h: Meyer-Wallach uses it too)

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

// Dense 4x4 on group bits (T0, T1) of the register tile, row = 2 * bit[T0] + bit[T1] (lds_apply's LK_2Q rule):
// four 4-vectors per work item.  The matrix comes straight from the sample's matrix row (wave-uniform address).
template <int T0, int T1>
__device__ __forceinline__ void reg_2q(float2 (&a)[16], const float *__restrict__ mm) {
  float2 M[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) M[i] = make_float2(mm[2 * i], mm[2 * i + 1]);
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if (c & ((1 << T0) | (1 << T1))) continue;
    const float2 a0 = a[c], a1 = a[c | (1 << T1)], a2 = a[c | (1 << T0)], a3 = a[c | (1 << T0) | (1 << T1)];
    float2 r[4];
#pragma unroll
    for (int row = 0; row < 4; ++row) {
      float2 acc = cmul(M[row * 4 + 0], a0);
      acc = cfma(M[row * 4 + 1], a1, acc);
      acc = cfma(M[row * 4 + 2], a2, acc);
      acc = cfma(M[row * 4 + 3], a3, acc);
      r[row] = acc;
    }
    a[c] = r[0];
    a[c | (1 << T1)] = r[1];
    a[c | (1 << T0)] = r[2];
    a[c | (1 << T0) | (1 << T1)] = r[3];
  }
  // (keeps the twelve instantiations apart: without it the optimiser sinks their identical tails out of the
  // dispatch switch, indexes a[] with the merged case number and the register tile lands in scratch)
  asm volatile("; reg_2q %0 %1" ::"n"(T0), "n"(T1));
}
__device__ __forceinline__ void reg_2q_dispatch(float2 (&a)[16], const float *__restrict__ mm, int t0, int t1) {
  switch (t0 * 4 + t1) {
    case 1: reg_2q<0, 1>(a, mm); break;
    case 2: reg_2q<0, 2>(a, mm); break;
    case 3: reg_2q<0, 3>(a, mm); break;
    case 4: reg_2q<1, 0>(a, mm); break;
    case 6: reg_2q<1, 2>(a, mm); break;
    case 7: reg_2q<1, 3>(a, mm); break;
    case 8: reg_2q<2, 0>(a, mm); break;
    case 9: reg_2q<2, 1>(a, mm); break;
    case 11: reg_2q<2, 3>(a, mm); break;
    case 12: reg_2q<3, 0>(a, mm); break;
    case 13: reg_2q<3, 1>(a, mm); break;
    default: reg_2q<3, 2>(a, mm); break;
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
// TWOQ: the run may hold uncontrolled LK_2Q ops as well (GK_REG4X).
template <bool SLOTS, bool TWOQ = false>
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
      if (TWOQ && op.kind == LK_2Q) {
        // (the descriptor came out of LDS: tell the compiler it is wave-uniform, so that the 4x4 matrix is
        // 32 scalar loads into SGPRs, not 32 VGPRs per lane)
        const uint32_t moff = (uint32_t)__builtin_amdgcn_readfirstlane((int)op.mat_off);
        const int tt = __builtin_amdgcn_readfirstlane((int)op.t0 * 4 + (int)op.t1);
        reg_2q_dispatch(a, mrow + moff, tt >> 2, tt & 3);
        continue;
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
  int shift;    // > 0: the tile is positions [shift, shift + T) (Stage::shift): amplitude j lives at j << shift
  int mw_lean;  // TM_STORE_MW of a tiled state: no cross terms for local bits 0..3 (= positions 0..3, which every
                // later read holds too: k_mw_read_later<.., LOW> reports them), populations out of the second gather
  int8_t tile_bits[QMLE_MAX_QUBITS];
  int8_t outer_bits[QMLE_MAX_QUBITS];
  uint32_t obs_mask[QMLE_MAX_QUBITS];  // per observable: bit p set <=> Z on bit position p
  uint16_t obs_local[QMLE_MAX_QUBITS]; // the same restricted to the tile, in LOCAL bits (bit j <=> Z on tile_bits[j])
  uint32_t obs_outer[QMLE_MAX_QUBITS]; // ... and to the outer positions, in TILE-INDEX bits (bit i <=> Z on outer_bits[i])
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
    } else if (DENSE4 && g.kind == GK_REG4X) {
      const uint32_t gm = (1u << g.bits[0]) | (1u << g.bits[1]) | (1u << g.bits[2]) | (1u << g.bits[3]);
      if (a.slots_in_lds) lds_apply_group<true, true>(s, T, g, a.ops, mrow, slots, a.op_begin, z & ~gm);
      else lds_apply_group<false, true>(s, T, g, a.ops, mrow, slots, a.op_begin, z & ~gm);
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
template <bool RAW, bool PARTIAL_ONLY = false,  // PARTIAL_ONLY: a.meas is TM_EXPVAL_PARTIAL (k_tile2's
                                                 // multi-tile instantiation keeps its register budget)
          bool SHIFTED = false>                  // the caller may be handed a Stage::shift tile (k_tile only)
__device__ __forceinline__ void tile_epilogue(const TileArgs &a, float2 *s, const uint32_t *lut,
                                              float *red, uint32_t tile, uint32_t n_tiles, int b,
                                              uint64_t base, int qsrc_of_thread = -1) {
  const int T = a.T, L = a.L;
  const int tid = threadIdx.x, nt = blockDim.x;
  const size_t D = (size_t)1 << a.n;
  const uint32_t half = 1u << (T - 1);
  const uint32_t lowmask = (1u << L) - 1u;
  float2 *st = a.states + (size_t)b * D;
  if (!PARTIAL_ONLY && (a.meas == TM_STORE || a.meas == TM_STORE_MW)) {
    if (SHIFTED && a.shift) {
      // the one tile per state of a top-first schedule: its amplitudes sit 2^shift apart, one 8-byte store each
      // (2^14 of them per state, behind the fill that wrote the zeros; launch_tile admits no other use)
      for (uint32_t j = tid; j < (1u << T); j += nt) st[(uint64_t)j << a.shift] = s[sw(j)];
    } else if ((half % (8u * nt)) == 0) {
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
      // Walsh-Hadamard transform of the work item's 16 probabilities over the 4 iteration bits (in registers):
      // w[i] = sum_it (-1)^{|it & i|} p(it).  An observable then needs ONE of the 16 (its iteration-bit mask),
      // signed by the parity of the lane index under its lane mask and summed over the wave: a wave-uniform pick,
      // one select and six DPP adds per observable -- eight observables per round of wave sums, nothing on the LDS
      // crossbar.  (Until round 5 the transform went on across the six lane bits -- 96 cross-lane exchanges + 16 LDS
      // stores per work item and tile for all 1024 parities of the wave, of which <= 32 were read: that epilogue
      // made the measuring pass of folded-CX plans LDS-bound, 47 % LDS busy with 22 % bank conflicts,
      // profiles/r05_deep_default_sq.txt.)
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
      // (through the kernel argument segment: indexing the by-value struct with a run-time index makes hipcc copy it
      // to scratch)
      const uint16_t QMLE_CONSTANT *ol =
          (const uint16_t QMLE_CONSTANT *)((const char QMLE_CONSTANT *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TileArgs, obs_local));
      const int n_obs = a.n_obs;
      tile_sync<RAW>();  // all amplitudes have been read: the tile buffer becomes scratch
      float *C = reinterpret_cast<float *>(s);  // [n_obs][nw]
      for (int k0 = 0; k0 < n_obs; k0 += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          v[k] = 0.f;
          if (k0 + k < n_obs) {  // (wave-uniform)
            const uint32_t lm = ol[k0 + k];
            float sel;
            switch (lm >> (T - 4)) {  // the observable's iteration-bit mask: wave-uniform
              case 0: sel = w[0]; break; case 1: sel = w[1]; break; case 2: sel = w[2]; break; case 3: sel = w[3]; break;
              case 4: sel = w[4]; break; case 5: sel = w[5]; break; case 6: sel = w[6]; break; case 7: sel = w[7]; break;
              case 8: sel = w[8]; break; case 9: sel = w[9]; break; case 10: sel = w[10]; break; case 11: sel = w[11]; break;
              case 12: sel = w[12]; break; case 13: sel = w[13]; break; case 14: sel = w[14]; break; default: sel = w[15]; break;
            }
            v[k] = (__popc((uint32_t)lane & lm & 63u) & 1) ? -sel : sel;
          }
        }
        wave_sums_dpp63(v);
        if (lane == kWave - 1) {
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (k0 + k < n_obs) C[(k0 + k) * nw + wv] = v[k];
        }
      }
      tile_sync<RAW>();
      if (tid < n_obs) {
        const uint32_t m = a.obs_mask[tid];
        const uint32_t mw = ((uint32_t)ol[tid] >> 6) & ((1u << (T - 10)) - 1u);
        uint32_t par = 0;
        for (int i = 0; i < a.n - T; ++i) par ^= ((m >> a.outer_bits[i]) & 1u) & ((tile >> i) & 1u);
        float r = 0.f;
        for (int v = 0; v < nw; ++v) {
          const float c = C[tid * nw + v];
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

// ---- Meyer-Wallach sums out of a finished tile (TM_STORE_MW / TM_MW_ONLY) --------------------
// The pass that produces the state has its 2^T amplitudes in LDS anyway: instead of re-reading them
// from HBM (the stand-alone first read of qmle_meyer_wallach), it reports what Tr rho_j^2 needs from
// this tile -- for every tile-local bit j the cross term c_j = sum_{bit_j = 0} psi_e conj(psi_{e + 2^j})
// and the signed population sum_e (-1)^{bit_j(e)} |psi_e|^2, plus the tile's total, from which the
// populations of every OUTER position follow (sign of a tile = a bit of its index).  One row of
// kMwFusedRow floats per tile (layout: tile_mw_finish_t).
// Work items: nt = 2^(T-4), 16 amplitudes per register gather, (T + 3) / 4 gathers over 4 local bits
// each (the last one overlaps its predecessor when T is not a multiple of 4); the populations come
// from the first gather (bits 0..3 in registers, bit j >= 4 = bit j - 4 of the thread index).
constexpr int kMwFusedRow = 48;

// the sums a work item carries across the tiles its workgroup walks (indexed with literals only)
struct MwAcc {
  v2f cr[14];
  float h0, h1, h2, h3, tot;
  float zw[4];  // the total signed by bit k of the tile's index inside the walk
};
__device__ __forceinline__ void tile_mw_clear(MwAcc &m) {
  static_for<14>([&](auto j) { m.cr[j] = (v2f){0.f, 0.f}; });
  m.h0 = m.h1 = m.h2 = m.h3 = m.tot = 0.f;
  static_for<4>([&](auto k) { m.zw[k] = 0.f; });
}

// one tile (in LDS, identity layout) into the work item's sums; `it` = index of the tile in the walk
// LEAN (tiled states, round 5): the gather over local bits 0..3 is dropped -- those are positions 0..3, which sit
// in the tile of every later read as well, and the later reads are HBM-bound with half of their vector issue
// slots free (profiles/r05_mw_sq_*.txt) while this pass is bound by its arithmetic: 16 LDS reads, 64 packed fmas
// and 8 of the 39 reduced values less per work item and tile.  The populations then come out of the second
// gather (register bits = local bits 4..7, thread bits = local bits 0..3 and 8..).
template <int T, bool LEAN>
__device__ __forceinline__ void tile_mw_accumulate_t(uint32_t sbo, uint32_t tid, uint32_t it, MwAcc &m) {
  static_assert(T >= 10 && T <= 14, "16 amplitudes per work item, at most 1024 work items");
  constexpr int G = (T + 3) / 4;
  constexpr int KPOP = LEAN ? 1 : 0;  // the gather the populations are taken from
  uint32_t tg = tid;
  asm volatile("" : "+v"(tg));  // (the gather addresses are not worth keeping across the gate loop)
  static_for<G>([&](auto k) {
    if constexpr (LEAN && (int)k == 0) return;
    constexpr int B = (4 * (int)k <= T - 4) ? 4 * (int)k : T - 4;
    constexpr int first_new = 4 * (int)k - B;  // bits [B, B + first_new) were reported by the previous gather
    const uint32_t bs = (sw(ins0(ins0(ins0(ins0(tg, B), B + 1), B + 2), B + 3)) << 3) + sbo;
    v2f r[16];
    static_for<16>([&](auto c) {
      const u64 x = lds_ld64(bs ^ (sw((uint32_t)c << B) << 3));
      r[c] = (v2f){__uint_as_float((uint32_t)x), __uint_as_float((uint32_t)(x >> 32))};
    });
    static_for<4>([&](auto t) {
      if constexpr ((int)t >= first_new) {
        static_for<8>([&](auto pq) {
          constexpr int lowm = (1 << t) - 1;
          constexpr int c = (((int)pq & ~lowm) << 1) | ((int)pq & lowm);
          mw_cross(m.cr[B + (int)t], r[c], r[c | (1 << t)]);
        });
      }
    });
    if constexpr ((int)k == KPOP) {  // populations: pruned Walsh-Hadamard butterfly over the four register bits
      float pr[16], s1[8], s2[4], s3[2], h0 = 0.f, h1 = 0.f, h2 = 0.f;
      static_for<16>([&](auto c) { const v2f q = r[c] * r[c]; pr[c] = q.x + q.y; });
      static_for<8>([&](auto i) { s1[i] = pr[2 * i] + pr[2 * i + 1]; h0 += pr[2 * i] - pr[2 * i + 1]; });
      static_for<4>([&](auto i) { s2[i] = s1[2 * i] + s1[2 * i + 1]; h1 += s1[2 * i] - s1[2 * i + 1]; });
      static_for<2>([&](auto i) { s3[i] = s2[2 * i] + s2[2 * i + 1]; h2 += s2[2 * i] - s2[2 * i + 1]; });
      const float tt = s3[0] + s3[1];
      m.h0 += h0; m.h1 += h1; m.h2 += h2; m.h3 += s3[0] - s3[1];
      m.tot += tt;
      static_for<4>([&](auto b) { m.zw[b] += __uint_as_float(__float_as_uint(tt) ^ (((it >> b) & 1u) << 31)); });
    }
  });
}
// one reduction and one row per workgroup: [0, 2T) cross terms, [2T, 3T) signed populations of the
// local bits, [3T] total, [3T + 1 + k] total signed by walk bit k.  `red`: LDS scratch (may be the tile:
// the first barrier below makes sure every gather has read it).  (One row per WAVE straight from lane 63
// -- no barrier, no scratch -- was measured too: the pass no faster, 8 x the rows for the purity kernel.)
template <int T, bool LEAN>
__device__ __forceinline__ void tile_mw_finish_t(const MwAcc &m, uint32_t tid, float *red, float *row_out) {
  // per-lane values: cross terms of the reported local bits [C0, T), then h0..h3 (populations of the four register
  // bits of the population gather = local bits [H0, H0 + 4)), the total, the total signed by thread bits 0..5, and
  // the walk-signed totals
  constexpr int C0 = LEAN ? 4 : 0, H0 = LEAN ? 4 : 0, NC = 2 * (T - C0);
  constexpr int NV = NC + 15, NW = 1 << (T - 10);
  float v[NV];
  static_for<T - C0>([&](auto j) { v[2 * j] = m.cr[C0 + j].x; v[2 * j + 1] = m.cr[C0 + j].y; });
  v[NC] = m.h0; v[NC + 1] = m.h1; v[NC + 2] = m.h2; v[NC + 3] = m.h3; v[NC + 4] = m.tot;
  static_for<6>([&](auto b) { v[NC + 5 + b] = ((tid >> b) & 1u) ? -m.tot : m.tot; });
  static_for<4>([&](auto k) { v[NC + 11 + k] = m.zw[k]; });
  wave_sums_dpp63(v);
  __syncthreads();  // every gather (and the store before it) has read the tile: it becomes scratch
  const uint32_t lane = tid & (kWave - 1), w = tid / kWave;
  if (lane == kWave - 1) static_for<NV>([&](auto i) { red[w * NV + i] = v[i]; });
  __syncthreads();
  if (tid < 3u * T + 5u) {
    // column of the per-wave sums this row entry reads (-1: not reported, 0), or the totals signed by a wave-index bit
    int col, wbit = -1;
    if (tid < 2u * T) col = (int)tid >= 2 * C0 ? (int)tid - 2 * C0 : -1;  // cross terms
    else if (tid < 3u * T) {
      const int j = (int)tid - 2 * T;                      // population of local bit j
      if (j >= H0 && j < H0 + 4) col = NC + (j - H0);      // a register bit of the population gather
      else {
        const int tb = j < H0 ? j : j - 4;                 // else bit tb of the thread index
        if (tb < 6) col = NC + 5 + tb;
        else { col = NC + 4; wbit = tb - 6; }
      }
    } else if (tid == 3u * T) col = NC + 4;                // the total
    else col = NC + 11 + ((int)tid - 3 * T - 1);           // walk-bit signed totals
    float s = 0.f;
    if (col >= 0) {
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const float x = red[i * NV + col];
        s += (wbit >= 0 && ((i >> wbit) & 1)) ? -x : x;
      }
    }
    row_out[tid] = s;
  }
}
// (T and lean are wave-uniform: kernel arguments)
__device__ __forceinline__ void tile_mw_accumulate(uint32_t sbo, int T, uint32_t tid, uint32_t it, MwAcc &m, bool lean = false) {
  if (lean) {
    switch (T) {
      case 10: tile_mw_accumulate_t<10, true>(sbo, tid, it, m); break;
      case 11: tile_mw_accumulate_t<11, true>(sbo, tid, it, m); break;
      case 12: tile_mw_accumulate_t<12, true>(sbo, tid, it, m); break;
      case 13: tile_mw_accumulate_t<13, true>(sbo, tid, it, m); break;
      default: tile_mw_accumulate_t<14, true>(sbo, tid, it, m); break;
    }
    return;
  }
  switch (T) {
    case 10: tile_mw_accumulate_t<10, false>(sbo, tid, it, m); break;
    case 11: tile_mw_accumulate_t<11, false>(sbo, tid, it, m); break;
    case 12: tile_mw_accumulate_t<12, false>(sbo, tid, it, m); break;
    case 13: tile_mw_accumulate_t<13, false>(sbo, tid, it, m); break;
    default: tile_mw_accumulate_t<14, false>(sbo, tid, it, m); break;
  }
}
__device__ __forceinline__ void tile_mw_finish(const MwAcc &m, int T, uint32_t tid, float *red, float *row_out, bool lean = false) {
  if (lean) {
    switch (T) {
      case 10: tile_mw_finish_t<10, true>(m, tid, red, row_out); break;
      case 11: tile_mw_finish_t<11, true>(m, tid, red, row_out); break;
      case 12: tile_mw_finish_t<12, true>(m, tid, red, row_out); break;
      case 13: tile_mw_finish_t<13, true>(m, tid, red, row_out); break;
      default: tile_mw_finish_t<14, true>(m, tid, red, row_out); break;
    }
    return;
  }
  switch (T) {
    case 10: tile_mw_finish_t<10, false>(m, tid, red, row_out); break;
    case 11: tile_mw_finish_t<11, false>(m, tid, red, row_out); break;
    case 12: tile_mw_finish_t<12, false>(m, tid, red, row_out); break;
    case 13: tile_mw_finish_t<13, false>(m, tid, red, row_out); break;
    default: tile_mw_finish_t<14, false>(m, tid, red, row_out); break;
  }
}
// one tile, one row
__device__ __forceinline__ void tile_mw_row(uint32_t sbo, int T, uint32_t tid, float *red, float *row_out, bool lean = false) {
  MwAcc m;
  tile_mw_clear(m);
  tile_mw_accumulate(sbo, T, tid, 0u, m, lean);
  tile_mw_finish(m, T, tid, red, row_out, lean);
}

// Host side: the kernel arguments of stage `st` (positions, known-zero masks, <Z> row sources).
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
  a.shift = st.shift;
  a.n_slots = p->n_slots;
  a.init_zero = init_zero ? 1 : 0;
  a.meas = meas;
  a.n_obs = n_obs;
  std::memcpy(a.tile_bits, st.tile_bits, sizeof(a.tile_bits));
  std::memcpy(a.outer_bits, st.outer_bits, sizeof(a.outer_bits));
  if (obs_masks) {
    std::memcpy(a.obs_mask, obs_masks, (size_t)n_obs * sizeof(uint32_t));
    for (int k = 0; k < n_obs && k < QMLE_MAX_QUBITS; ++k) {
      uint32_t lm = 0;
      for (int j = 0; j < st.T && j < 16; ++j) lm |= ((obs_masks[k] >> st.tile_bits[j]) & 1u) << j;
      a.obs_local[k] = (uint16_t)lm;
      uint32_t om = 0;
      for (int i = 0; i < p->n - st.T && i < 32; ++i) om |= ((obs_masks[k] >> st.outer_bits[i]) & 1u) << i;
      a.obs_outer[k] = om;
    }
  }
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

}  // namespace
