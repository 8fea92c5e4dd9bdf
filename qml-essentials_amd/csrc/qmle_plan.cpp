// Host-side plan compiler: tape validation, lowering to controlled-2x2 / 4x4
// primitives, commutation-aware merging of 1-qubit gates, and greedy scheduling
// of the lowered ops into HBM passes ("stages").  No HIP calls in this file, so
// plans can be built and inspected on a machine without a GPU.
//
// What is being replaced: jax traces the circuit once and XLA emits one
// out-of-place einsum per gate (qml_essentials/simulation.py:91-104).  Here the
// tape is compiled into a few passes; each pass stages a 2^T-amplitude tile in
// LDS and applies every gate whose wires fall inside the tile.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <sstream>

#include "qmle_internal.h"

namespace qmle {

namespace {

struct OpInfo {
  int n_wires;
  int n_params;
  bool has_const;
};

bool op_info(int opcode, OpInfo *info) {
  switch (opcode) {
    case QMLE_OP_ID: case QMLE_OP_X: case QMLE_OP_Y: case QMLE_OP_Z:
    case QMLE_OP_H: case QMLE_OP_S:
      *info = {1, 0, false}; return true;
    case QMLE_OP_RX: case QMLE_OP_RY: case QMLE_OP_RZ:
      *info = {1, 1, false}; return true;
    case QMLE_OP_ROT:
      *info = {1, 3, false}; return true;
    case QMLE_OP_CX: case QMLE_OP_CY: case QMLE_OP_CZ: case QMLE_OP_SWAP:
      *info = {2, 0, false}; return true;
    case QMLE_OP_CRX: case QMLE_OP_CRY: case QMLE_OP_CRZ: case QMLE_OP_CPHASE:
    case QMLE_OP_RXX: case QMLE_OP_RYY: case QMLE_OP_RZZ: case QMLE_OP_RZX:
      *info = {2, 1, false}; return true;
    case QMLE_OP_CCX: case QMLE_OP_CSWAP:
      *info = {3, 0, false}; return true;
    case QMLE_OP_MAT1:
      *info = {1, 0, true}; return true;
    case QMLE_OP_MAT2:
      *info = {2, 0, true}; return true;
    case QMLE_OP_DIAG_ALL:
      *info = {-1, 1, true}; return true;
    case QMLE_OP_MAT4:
      *info = {4, 0, true}; return true;
    default:
      return false;
  }
}

bool is_diag_opcode(int opcode) {
  switch (opcode) {
    case QMLE_OP_ID: case QMLE_OP_Z: case QMLE_OP_S: case QMLE_OP_RZ:
    case QMLE_OP_CZ: case QMLE_OP_CRZ: case QMLE_OP_CPHASE: case QMLE_OP_RZZ:
      return true;
    default:
      return false;
  }
}

inline uint64_t bit(int p) { return 1ull << p; }

uint64_t op_mask(const LoweredOp &o, int n) {
  if (o.kind == LK_DIAG_ALL) return n >= 64 ? ~0ull : (bit(n) - 1);
  uint64_t m = bit(o.t0);
  if (o.t1 >= 0) m |= bit(o.t1);
  if (o.c0 >= 0) m |= bit(o.c0);
  if (o.c1 >= 0) m |= bit(o.c1);
  return m;
}

}  // namespace

// SURVEY.md 8-d / BASELINE.md section 3: algorithmic bytes per state of one
// reference gate (complex64, D = 2^n).
double algo_bytes(const qmle_op &op, int n) {
  const double D = (double)(1ull << n);
  switch (op.opcode) {
    case QMLE_OP_ID: return 0.0;
    case QMLE_OP_CX: case QMLE_OP_CY: case QMLE_OP_CRX: case QMLE_OP_CRY:
    case QMLE_OP_CRZ: return 8.0 * D;
    case QMLE_OP_CZ: case QMLE_OP_CPHASE: case QMLE_OP_CCX: case QMLE_OP_CSWAP:
      return 4.0 * D;
    default: return 16.0 * D;
  }
}

constexpr bool kSmallLastTileDefault = true;  // (measured: profiles/r05_small_last_tile_ab.txt)

// Partition the ops of one tile stage into register-tile groups (<= 4 tile-local bits
// per group, dependency order preserved: an op may only move ahead of ops it shares no
// bit with).  Rewrites dev_ops[st.op_begin, st.op_end) into group order.
static void group_stage_ops(qmle_plan *p, Stage &st) {
  const int nops = st.op_end - st.op_begin;
  st.grp_begin = (int)p->op_groups.size();
  std::vector<LoweredOp> src(p->dev_ops.begin() + st.op_begin, p->dev_ops.begin() + st.op_end);
  std::vector<int> src_id(p->dev_src.begin() + st.op_begin, p->dev_src.begin() + st.op_end);
  std::vector<LoweredOp> out;
  std::vector<int> out_id;
  out.reserve(nops);
  std::vector<char> done(nops, 0);
  const bool regs_ok = st.T >= 4 && !(p->flags & QMLE_PLAN_NO_REGTILE);
  // (round 5: an uncontrolled dense 4x4 joins a register-tile run as well -- the noisy model's superoperators sit
  // between every pair of gates and cost an LDS sweep each otherwise; QMLE_NO_REG2Q=1: A/B.  Plans that keep every
  // gate its own operator (QMLE_PLAN_NO_MERGE: the adjoint sweep's, whose tile kernel knows GK_REG4 only) do not)
  static const bool no_reg2q = std::getenv("QMLE_NO_REG2Q") != nullptr;
  const bool reg2q = regs_ok && !no_reg2q && !(p->flags & QMLE_PLAN_NO_MERGE);
  auto groupable = [&](const LoweredOp &o) {
    return regs_ok && ((o.kind == LK_1Q && o.nc <= 1) || (reg2q && o.kind == LK_2Q && o.nc == 0));
  };
  auto mask_of = [&](const LoweredOp &o) -> uint32_t {
    if (o.kind == LK_DIAG_ALL) return st.T >= 32 ? ~0u : ((1u << st.T) - 1u);
    uint32_t m = 1u << o.t0;
    if (o.t1 >= 0) m |= 1u << o.t1;
    if (o.c0 >= 0) m |= 1u << o.c0;
    if (o.c1 >= 0) m |= 1u << o.c1;
    return m;
  };
  int n_done = 0;
  while (n_done < nops) {
    int first = 0;
    while (done[first]) ++first;
    OpGroup g{};
    g.op_begin = (uint32_t)(st.op_begin + out.size());
    if (src[first].kind == LK_4Q && st.T >= 4) {
      // gather order = ascending tile-local bit; permute the 16x16 matrix (given in wire
      // order, first wire = MSB) into that order once, on the host
      LoweredOp o = src[first];
      const int pos4[4] = {o.t0, o.t1, o.c0, o.c1};  // wire order, MSB first
      int sorted[4] = {pos4[0], pos4[1], pos4[2], pos4[3]};
      std::sort(sorted, sorted + 4);
      int rank_of[4];  // matrix bit (3 - j) of wire j  ->  gather bit = rank of its position
      for (int j = 0; j < 4; ++j)
        rank_of[j] = (int)(std::find(sorted, sorted + 4, pos4[j]) - sorted);
      auto to_wire_index = [&](int c) {  // gather index c -> matrix index in wire order
        int m = 0;
        for (int j = 0; j < 4; ++j)
          if (c & (1 << rank_of[j])) m |= 1 << (3 - j);
        return m;
      };
      const size_t src_off = o.mat_off, dst_off = p->consts.size();
      p->consts.resize(dst_off + 512);
      for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) {
          const int mr = to_wire_index(r), mc = to_wire_index(c);
          p->consts[dst_off + 2 * (r * 16 + c)] = p->consts[src_off + 2 * (mr * 16 + mc)];
          p->consts[dst_off + 2 * (r * 16 + c) + 1] = p->consts[src_off + 2 * (mr * 16 + mc) + 1];
        }
      o.mat_off = (uint32_t)dst_off;
      g.kind = GK_DENSE4;
      g.n_ops = 1;
      for (int j = 0; j < 4; ++j) g.bits[j] = (uint8_t)sorted[j];
      out.push_back(o);
      out_id.push_back(src_id[first]);
      done[first] = 1;
      ++n_done;
      p->op_groups.push_back(g);
      continue;
    }
    if (!groupable(src[first])) {
      g.kind = GK_SWEEP;
      g.n_ops = 1;
      out.push_back(src[first]);
      out_id.push_back(src_id[first]);
      done[first] = 1;
      ++n_done;
      p->op_groups.push_back(g);
      continue;
    }
    g.kind = GK_REG4;
    uint32_t G = 0, blocked = 0;
    std::vector<int> mem;
    for (int i = first; i < nops && mem.size() < 255; ++i) {
      if (done[i]) continue;
      const uint32_t m = mask_of(src[i]);
      if ((m & blocked) || !groupable(src[i])) {
        blocked |= m;
      } else if (__builtin_popcount(G | m) <= 4) {
        G |= m;
        mem.push_back(i);
      } else {
        blocked |= m;
      }
    }
    // pad to 4 bits: prefer positions >= 5 (bank-conflict-free gathers), then anything free
    for (int b = 5; b < st.T && __builtin_popcount(G) < 4; ++b) G |= 1u << b;
    for (int b = 0; b < st.T && __builtin_popcount(G) < 4; ++b) G |= 1u << b;
    int8_t local_of[32];
    int k = 0;
    for (int b = 0; b < st.T; ++b) {
      local_of[b] = -1;
      if (G & (1u << b)) { g.bits[k] = (uint8_t)b; local_of[b] = (int8_t)k++; }
    }
    g.n_ops = (uint8_t)mem.size();
    for (int i : mem) {
      LoweredOp o = src[i];
      o.t0 = local_of[(int)o.t0];
      if (o.kind == LK_2Q) {
        o.t1 = local_of[(int)o.t1];
        g.kind = GK_REG4X;
      }
      if (o.c0 >= 0) o.c0 = local_of[(int)o.c0];
      out.push_back(o);
      out_id.push_back(src_id[i]);
      done[i] = 1;
      ++n_done;
    }
    p->op_groups.push_back(g);
  }
  std::copy(out.begin(), out.end(), p->dev_ops.begin() + st.op_begin);
  std::copy(out_id.begin(), out_id.end(), p->dev_src.begin() + st.op_begin);
  st.grp_end = (int)p->op_groups.size();
}

// ---- fast tile path: perm-fused register-tile groups (Group2) --------------------------------
// `src`: the stage's ops in a valid execution order, TILE-local bit positions.  Greedy like
// group_stage_ops, except that X / CX never cost a group of their own:
//   * one that touches no bit of the group being formed (and no blocked bit) is applied to the
//     layout map at once -- the group's gather table already sees it;
//   * one at the END of a group's op list is peeled off and applied to the layout map after the
//     group (its scatter stays in place);
//   * only an X / CX sandwiched between the group's dense gates runs in registers (8 moves).
// Layout map: logical tile index e lives at physical slot L(e) = XOR_{j in e} Lcol[j] ^ Lconst.
static inline uint32_t swz(uint32_t e) { return e ^ (((e >> 5) & 15u) << 1); }  // = sw() in qmle_dev.h

static void build_fast_groups(qmle_plan *p, Stage &st, const std::vector<LoweredOp> &src,
                              uint32_t zin_local) {
  st.fast_ok = false;
  st.fast_begin = st.fast_end = (int)p->groups2.size();
  const int T = st.T;
  if (T < kFastMinT || T > kFastMaxT || (p->flags & QMLE_PLAN_NO_REGTILE)) return;
  for (const LoweredOp &o : src)
    if (o.kind != LK_1Q || o.nc > 1) return;
  const int nops = (int)src.size();
  const uint32_t nt = 1u << (T - 4);
  const size_t ops2_mark = p->ops2.size(), tbl_mark = p->tbl2.size();
  uint32_t Lcol[16], Lconst = 0;
  for (int j = 0; j < T; ++j) Lcol[j] = 1u << j;
  auto L_of = [&](uint32_t e) {
    uint32_t v = Lconst;
    for (int j = 0; j < T; ++j)
      if (e & (1u << j)) v ^= Lcol[j];
    return v;
  };
  auto L_is_identity = [&]() {
    if (Lconst) return false;
    for (int j = 0; j < T; ++j)
      if (Lcol[j] != (1u << j)) return false;
    return true;
  };
  // M: logical index in the frame of the last emitted group -> logical index now
  uint32_t Mcol[16], Mconst = 0;
  auto M_reset = [&]() { for (int j = 0; j < T; ++j) Mcol[j] = 1u << j; Mconst = 0; };
  M_reset();
  auto is_perm = [](const LoweredOp &o) { return (o.flags & LF_PERMX) != 0; };
  // known-zero tile-local bits along the stage's execution (runs from |0..0>, Stage::zero_in):
  // a gate that is not diagonal takes its target out of the set, controls stay.  A work item
  // whose base index meets the set outside its group's bits holds 16 exact zeros: bit 0 of its
  // table entry says so (the kernel honours it only in runs that track known zeros).
  uint32_t Z = zin_local;
  auto apply_perm = [&](const LoweredOp &o) {
    Z &= ~(1u << o.t0);
    const int t = o.t0;
    if (o.nc == 0) {
      Lconst ^= Lcol[t];
      Mconst ^= 1u << t;
    } else {
      const int c = o.c0;
      Lcol[c] ^= Lcol[t];
      for (int j = 0; j < T; ++j) Mcol[j] ^= ((Mcol[j] >> c) & 1u) << t;
      Mconst ^= ((Mconst >> c) & 1u) << t;
    }
  };
  auto mask_of = [](const LoweredOp &o) {
    uint32_t m = 1u << o.t0;
    if (o.c0 >= 0) m |= 1u << o.c0;
    return m;
  };
  // Thread index bit k of a group's work items -> tile-local position pos_of[k] (outside the group's
  // bits).  Any bijection is correct; WHICH one decides the LDS bank conflicts of the group's 32
  // accesses: slot(thread, c) = P(thread) ^ Q(c) with P linear over GF(2), so a ds_write_b64's 16-lane
  // groups are conflict-free iff thread bits 0..3 land on columns that are independent in slot bits
  // 0..3 (16 slots = 32 banks), and a ds_read_b64's 32-lane groups iff bits 0..4 are independent in
  // slot bits 0..4 (MI355X_MICROARCH.md, LDS table).  Ascending order -- the round-2 choice -- left
  // 33-38 % of the LDS cycles of the whole-state kernels to conflicts (SQ_LDS_BANK_CONFLICT,
  // profiles/r04_ws_sq_*.txt).  Only the lane bits are permuted: the wave index keeps the highest
  // positions, where the known zeros of a run from |0..0> sit (whole waves of idle work items skip).
  int pos_of[16];
  static const bool no_bank_perm = std::getenv("QMLE_NO_BANK_PERM") != nullptr;
  auto choose_thread_bits = [&](uint32_t G, bool keep_order) {
    int freep[16], nf = 0;
    for (int j = 0; j < T; ++j)
      if (!(G & (1u << j))) freep[nf++] = j;
    for (int k = 0; k < nf; ++k) pos_of[k] = freep[k];
    const int lanes = nf < 6 ? nf : 6;
    if (keep_order || no_bank_perm || lanes < 2) return;
    uint32_t basis[8];
    int nb = 0;
    auto independent = [&](uint32_t v) {  // reduce v by the basis; keep it when something is left
      for (int i = 0; i < nb; ++i)
        if ((v ^ basis[i]) < v) v ^= basis[i];
      if (!v) return false;
      basis[nb++] = v;
      for (int i = nb - 1; i > 0 && basis[i] > basis[i - 1]; --i) std::swap(basis[i], basis[i - 1]);
      return true;
    };
    bool used[16] = {};
    int order[16], no = 0;
    for (int k = 0; k < lanes && no < 4; ++k)  // four columns independent in slot bits 0..3
      if (independent(swz(Lcol[freep[k]]) & 0xFu)) { order[no++] = k; used[k] = true; }
    nb = 0;
    for (int i = 0; i < no; ++i) (void)independent(swz(Lcol[freep[order[i]]]) & 0x1Fu);
    const int first4 = no;
    for (int k = 0; k < lanes && no < first4 + 1; ++k)  // a fifth, independent in slot bits 0..4
      if (!used[k] && independent(swz(Lcol[freep[k]]) & 0x1Fu)) { order[no++] = k; used[k] = true; }
    // fewer than four independent columns exist: the group's own bits own those banks -- fill up
    // (thread bits 4 / 5 only choose the lane group of a write, bit 5 that of a read)
    for (int k = 0; k < lanes; ++k)
      if (!used[k]) order[no++] = k;
    for (int k = 0; k < lanes; ++k) pos_of[k] = freep[order[k]];
  };
  auto deposit = [&](uint32_t t, uint32_t G) {  // bits of t into the positions outside G
    (void)G;
    uint32_t e = 0;
    for (int k = 0; k < T - 4; ++k) e |= ((t >> k) & 1u) << pos_of[k];
    return e;
  };
  auto emit_tables = [&](Group2 &g, uint32_t G) {
    int gb[4], k = 0;
    for (int j = 0; j < T; ++j)
      if (G & (1u << j)) gb[k++] = j;
    // (the idle-work-item flags below are only read by tiled runs that track known zeros)
    choose_thread_bits(G, (Z & ~G) != 0 && T < p->n && !(p->flags & (QMLE_PLAN_NO_SPARSE | QMLE_PLAN_PREFETCH)));
    g.tbl = (uint32_t)p->tbl2.size();
    for (uint32_t t = 0; t < nt; ++t) {
      const uint32_t e = deposit(t, G);
      p->tbl2.push_back((swz(L_of(e)) << 3) | ((e & Z & ~G) ? 1u : 0u));
    }
    for (int c = 0; c < 16; ++c) {
      uint32_t v = 0;
      for (int j = 0; j < 4; ++j)
        if (c & (1 << j)) v ^= Lcol[gb[j]];
      g.off[c] = swz(v) << 3;
    }
  };
  std::vector<char> done(nops, 0);
  int n_done = 0;
  int last_group = -1;
  uint32_t last_G = 0;
  while (n_done < nops) {
    uint32_t G = 0, touched = 0, blocked = 0;
    std::vector<int> mem, after;  // `after`: X / CX applied to the layout behind the group
    for (int i = 0; i < nops; ++i) {
      if (done[i]) continue;
      const uint32_t m = mask_of(src[i]);
      if (m & blocked) { blocked |= m; continue; }
      if (is_perm(src[i])) {
        if (!(m & touched)) {  // independent of the group: layout only, before the group
          apply_perm(src[i]);
          done[i] = 1;
          ++n_done;
        } else if ((m & ~G) == 0 && mem.size() < 4000) {
          mem.push_back(i);  // inside the group's bits: stays in registers unless peeled below
        } else {             // would cost the group a bit position: behind the group instead
          after.push_back(i);
          blocked |= m;
        }
        continue;
      }
      if (__builtin_popcount(G | m) <= 4 && mem.size() < 4000) {
        G |= m;
        touched |= m;
        mem.push_back(i);
      } else {
        blocked |= m;
      }
    }
    if (mem.empty()) continue;  // only layout changes were left (`after` needs a member: empty too)
    {  // an X / CX that commutes with every later member of the group leaves it for the layout
      uint32_t later = 0;
      std::vector<int> keep;
      for (size_t k = mem.size(); k-- > 0;) {
        const int i = mem[k];
        const uint32_t m = mask_of(src[i]);
        if (is_perm(src[i]) && !(m & later)) {
          after.push_back(i);
        } else {
          later |= m;
          keep.insert(keep.begin(), i);
        }
      }
      mem.swap(keep);
      std::sort(after.begin(), after.end());
    }
    const std::vector<int> &trailing = after;
    G = 0;
    for (int i : mem) G |= mask_of(src[i]);
    for (int b = 5; b < T && __builtin_popcount(G) < 4; ++b) G |= 1u << b;
    for (int b = 0; b < T && __builtin_popcount(G) < 4; ++b) G |= 1u << b;
    int8_t local_of[16];
    {
      int k = 0;
      for (int b = 0; b < T; ++b) local_of[b] = (G & (1u << b)) ? (int8_t)k++ : (int8_t)-1;
    }
    Group2 g;
    std::memset(&g, 0, sizeof(g));
    g.op_begin = (uint32_t)p->ops2.size();
    g.n_ops = (uint16_t)mem.size();
    emit_tables(g, G);
    for (int i : mem)
      if (!(src[i].flags & LF_DIAG)) Z &= ~(1u << src[i].t0);
    for (int i : mem) {
      LoweredOp o = src[i];
      o.t0 = local_of[(int)o.t0];
      if (o.c0 >= 0) o.c0 = local_of[(int)o.c0];
      const int base = is_perm(o) ? (o.nc ? FC_CX : FC_X)
                       : (o.flags & LF_DIAG) ? (o.nc ? FC_CDIAG : FC_DIAG)
                                             : (o.nc ? FC_CDENSE : FC_DENSE);
      o.pad = (uint8_t)(base + (o.nc ? 3 * o.c0 + (o.t0 - (o.t0 > o.c0 ? 1 : 0)) : o.t0));
      p->ops2.push_back(o);
      done[i] = 1;
      ++n_done;
    }
    M_reset();
    for (int i : trailing) {
      apply_perm(src[i]);
      done[i] = 1;
      ++n_done;
    }
    last_group = (int)p->groups2.size();
    last_G = G;
    p->groups2.push_back(g);
  }
  // the epilogue (store / measure) reads the tile in the identity layout
  if (!L_is_identity() || last_group < 0) {
    uint32_t G = last_G;
    if (last_group < 0) {  // nothing but layout changes (or no gate at all): an empty group moves the data
      Group2 g;
      std::memset(&g, 0, sizeof(g));
      g.op_begin = (uint32_t)p->ops2.size();
      G = 0xFu << (T >= 9 ? 5 : 0);
      emit_tables(g, G);
      M_reset();
      last_group = (int)p->groups2.size();
      p->groups2.push_back(g);
    }
    Group2 &g = p->groups2[last_group];
    if (!L_is_identity()) {
      // a thread holds logical index e of the group's frame; its final logical index is M(e)
      int gb[4], k = 0;
      for (int j = 0; j < T; ++j)
        if (G & (1u << j)) gb[k++] = j;
      auto M_lin = [&](uint32_t e) {
        uint32_t v = 0;
        for (int j = 0; j < T; ++j)
          if (e & (1u << j)) v ^= Mcol[j];
        return v;
      };
      g.relayout = 1;
      g.tbl_out = (uint32_t)p->tbl2.size();
      for (uint32_t t = 0; t < nt; ++t) p->tbl2.push_back(swz(M_lin(deposit(t, G)) ^ Mconst) << 3);
      for (int c = 0; c < 16; ++c) {
        uint32_t e = 0;
        for (int j = 0; j < 4; ++j)
          if (c & (1 << j)) e |= 1u << gb[j];
        g.off_out[c] = swz(M_lin(e)) << 3;
      }
    }
  }
  (void)ops2_mark; (void)tbl_mark;
  // global byte offset of every lane's first float4 inside the tile: local index 2 t with its
  // bits deposited at the tile's global positions (bits below L are contiguous)
  st.fast_gtab = (uint32_t)p->tbl2.size();
  for (uint32_t t = 0; t < nt; ++t) {
    const uint32_t jl = 2u * t;
    uint32_t g = 0;
    for (int j = 0; j <= T - 4; ++j)
      if (jl & (1u << j)) g |= 1u << st.tile_bits[j];
    p->tbl2.push_back(g << 3);
  }
  st.fast_end = (int)p->groups2.size();
  st.fast_ok = true;
}

// ---- observable absorption ---------------------------------------------------------------
// Going backwards through the tape, a gate is absorbed iff it is a basis permutation with a
// LINEAR index map (CX, SWAP), a diagonal gate (phases drop out of |amplitude|^2) or the
// identity, and no later gate that stays in the circuit touches one of its wires.
static bool absorbable(uint16_t opcode) {
  switch (opcode) {
    case QMLE_OP_ID: case QMLE_OP_Z: case QMLE_OP_S: case QMLE_OP_RZ: case QMLE_OP_CZ:
    case QMLE_OP_CRZ: case QMLE_OP_CPHASE: case QMLE_OP_RZZ: case QMLE_OP_DIAG_ALL:
    case QMLE_OP_CX: case QMLE_OP_SWAP:
      return true;
    default:
      return false;
  }
}

void split_expval_tail(const std::vector<qmle_op> &ops, int n, std::vector<qmle_op> &kept,
                       std::vector<qmle_op> &absorbed) {
  kept.clear();
  absorbed.clear();
  std::vector<char> take(ops.size(), 0);
  uint32_t blocked = 0;  // wires a kept later gate acts on
  const uint32_t all = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
  for (size_t k = ops.size(); k-- > 0;) {
    const qmle_op &op = ops[k];
    uint32_t wires = 0;
    if (op.opcode == QMLE_OP_DIAG_ALL) wires = all;
    else
      for (int a = 0; a < 4 && op.wire[a] >= 0; ++a) wires |= 1u << op.wire[a];
    if (absorbable(op.opcode) && !(wires & blocked)) take[k] = 1;
    else blocked |= wires;
    if (blocked == all) break;  // nothing earlier can be absorbed
  }
  for (size_t k = 0; k < ops.size(); ++k) (take[k] ? absorbed : kept).push_back(ops[k]);
}

uint32_t pull_back_z(const std::vector<qmle_op> &absorbed, int wire) {
  uint32_t m = 1u << wire;
  for (size_t k = absorbed.size(); k-- > 0;) {
    const qmle_op &op = absorbed[k];
    if (op.opcode == QMLE_OP_CX) {            // Z_t -> Z_c Z_t, Z_c -> Z_c
      if (m & (1u << op.wire[1])) m ^= 1u << op.wire[0];
    } else if (op.opcode == QMLE_OP_SWAP) {
      const uint32_t a = (m >> op.wire[0]) & 1u, b = (m >> op.wire[1]) & 1u;
      if (a != b) m ^= (1u << op.wire[0]) | (1u << op.wire[1]);
    }
  }
  return m;
}

constexpr uint8_t kLoweredDead = 255;  // LoweredOp::kind of an operator merged away during lowering

int compile_plan(qmle_plan *p) {
  const int n = p->n;
  if (n < 1 || n > QMLE_MAX_QUBITS) return QMLE_ERR_INVALID_ARG;

  // ---- 1. validate + lower -------------------------------------------------
  std::vector<int> last_touch(n, -1);  // lowered index that last touched bit p
  p->lowered.clear(); p->lowered_src.clear(); p->build_ops.clear(); p->groups.clear();
  p->algo_bytes_per_state = 0;
  p->mat_floats = 0;
  const bool fuse = !(p->flags & QMLE_PLAN_NO_FUSION);
  const bool no_fusion_flag = !fuse;
  const size_t n_user_consts = p->consts.size();
  std::vector<std::vector<BuildOp>> group_ops;            // source gates per matrix
  std::vector<std::pair<uint32_t, uint32_t>> group_meta;  // (mat_off, dim)
  std::vector<int> group_of;                              // lowered index -> group

  for (size_t i = 0; i < p->ops.size(); ++i) {
    const qmle_op &op = p->ops[i];
    OpInfo info;
    if (!op_info(op.opcode, &info)) return QMLE_ERR_UNKNOWN_OP;
    int nw = 0;
    while (nw < 4 && op.wire[nw] >= 0) ++nw;
    if (op.opcode == QMLE_OP_DIAG_ALL) {
      // wires are implicitly 0..n-1 in order (operations.py:922-926)
      if (op.mat_off < 0 || (size_t)op.mat_off + (1ull << n) > p->consts.size())
        return QMLE_ERR_INVALID_ARG;
    } else {
      if (nw != info.n_wires) return QMLE_ERR_WIRE_COUNT;
      for (int a = 0; a < nw; ++a) {
        if (op.wire[a] >= n) return QMLE_ERR_WIRE_RANGE;
        for (int b = a + 1; b < nw; ++b)
          if (op.wire[a] == op.wire[b]) return QMLE_ERR_DUPLICATE_WIRES;
      }
      if (info.has_const) {
        const size_t need = op.opcode == QMLE_OP_MAT1 ? 8 : op.opcode == QMLE_OP_MAT2 ? 32 : 512;
        if (op.mat_off < 0 || (size_t)op.mat_off + need > p->consts.size())
          return QMLE_ERR_INVALID_ARG;
      }
    }
    for (int a = 0; a < info.n_params; ++a)
      if (op.slot[a] < 0 || op.slot[a] >= p->n_slots) return QMLE_ERR_SLOT_RANGE;

    p->algo_bytes_per_state += algo_bytes(op, n);
    if (op.opcode == QMLE_OP_ID) continue;  // identity: nothing to do

    auto pos = [n](int w) { return (int8_t)(n - 1 - w); };
    LoweredOp lo{};
    lo.t1 = lo.c0 = lo.c1 = -1;
    lo.slot = -1;
    lo.flags = is_diag_opcode(op.opcode) ? LF_DIAG : 0;
    if (op.opcode == QMLE_OP_X || op.opcode == QMLE_OP_CX || op.opcode == QMLE_OP_CCX)
      lo.flags |= LF_PERMX;
    if (op.opcode == QMLE_OP_CZ || op.opcode == QMLE_OP_CPHASE) lo.flags |= LF_PHASE;  // (operations.py:1100, 1171-1201)
    BuildOp bo{};
    bo.opcode = op.opcode;
    for (int a = 0; a < 3; ++a) bo.slot[a] = op.slot[a];
    bo.const_off = op.mat_off;

    if (op.opcode == QMLE_OP_DIAG_ALL) {
      lo.kind = LK_DIAG_ALL;
      lo.t0 = 0;
      lo.mat_off = (uint32_t)op.mat_off;
      lo.slot = op.slot[0];
      for (int b = 0; b < n; ++b) last_touch[b] = (int)p->lowered.size();
      group_of.resize(p->lowered.size() + 1, -1);
      p->lowered.push_back(lo);
      p->lowered_src.push_back({(int)i});
      continue;
    }

    if (op.opcode == QMLE_OP_MAT4) {
      lo.kind = LK_4Q;
      lo.t0 = pos(op.wire[0]); lo.t1 = pos(op.wire[1]);
      lo.c0 = pos(op.wire[2]); lo.c1 = pos(op.wire[3]);
      lo.nc = 0;
      lo.mat_off = (uint32_t)op.mat_off;  // const blob; permuted copy is made per stage
      const int idx4 = (int)p->lowered.size();
      const uint64_t m4 = op_mask(lo, n);
      for (int b = 0; b < n; ++b)
        if (m4 & bit(b)) last_touch[b] = idx4;
      group_of.resize(idx4 + 1, -1);
      p->lowered.push_back(lo);
      p->lowered_src.push_back({(int)i});
      continue;
    }

    switch (op.opcode) {
      case QMLE_OP_CX: case QMLE_OP_CY: case QMLE_OP_CZ: case QMLE_OP_CRX:
      case QMLE_OP_CRY: case QMLE_OP_CRZ: case QMLE_OP_CPHASE:
        lo.kind = LK_1Q; lo.nc = 1; lo.c0 = pos(op.wire[0]); lo.t0 = pos(op.wire[1]);
        break;
      case QMLE_OP_CCX:
        lo.kind = LK_1Q; lo.nc = 2; lo.c0 = pos(op.wire[0]); lo.c1 = pos(op.wire[1]);
        lo.t0 = pos(op.wire[2]);
        break;
      case QMLE_OP_CSWAP:
        lo.kind = LK_2Q; lo.nc = 1; lo.c0 = pos(op.wire[0]); lo.t0 = pos(op.wire[1]);
        lo.t1 = pos(op.wire[2]);
        break;
      case QMLE_OP_SWAP: case QMLE_OP_RXX: case QMLE_OP_RYY: case QMLE_OP_RZZ:
      case QMLE_OP_RZX: case QMLE_OP_MAT2:
        lo.kind = LK_2Q; lo.nc = 0; lo.t0 = pos(op.wire[0]); lo.t1 = pos(op.wire[1]);
        break;
      default:  // uncontrolled 1-qubit
        lo.kind = LK_1Q; lo.nc = 0; lo.t0 = pos(op.wire[0]);
        break;
    }

    // commutation-aware merge: an uncontrolled 1-q gate multiplies onto the
    // previous uncontrolled 1-q matrix on the same wire if nothing touched that
    // wire in between (gates on disjoint wires commute).
    if (fuse && !(p->flags & QMLE_PLAN_NO_MERGE) && lo.kind == LK_1Q && lo.nc == 0) {
      const int prev = last_touch[lo.t0];
      if (prev >= 0) {
        LoweredOp &pl = p->lowered[prev];
        if (pl.kind == LK_1Q && pl.nc == 0 && pl.t0 == lo.t0) {
          group_ops[group_of[prev]].push_back(bo);
          if (!(lo.flags & LF_DIAG)) pl.flags &= ~LF_DIAG;
          pl.flags &= ~LF_PERMX;  // a product of gates is a general 2x2
          p->lowered_src[prev].push_back((int)i);
          continue;
        }
      }
    }

    // Round 5: the same holds around an uncontrolled dense 4x4 (the Kraus superoperators of vec(rho) on [w, n + w],
    // two-qubit Pauli rotations): a 1-q gate on one of its wires multiplies onto it as U (x) I / I (x) U
    // (BuildOp::pad = 1 / 2), a 4x4 on the same ordered pair as a plain product, and a new 4x4 takes the
    // pending 1-q matrices of its two wires with it.  A noisy model's gate-channel-gate-channel run on one wire
    // (U (x) conj U, superoperator, ...) becomes ONE 4x4 per sample.  QMLE_NO_MERGE_2Q=1: A/B.
    static const bool no_merge_2q = std::getenv("QMLE_NO_MERGE_2Q") != nullptr;
    const bool merge_2q = fuse && !(p->flags & QMLE_PLAN_NO_MERGE) && !no_merge_2q;
    if (merge_2q && lo.kind == LK_1Q && lo.nc == 0) {
      const int prev = last_touch[lo.t0];
      if (prev >= 0) {
        LoweredOp &pl = p->lowered[prev];
        if (pl.kind == LK_2Q && pl.nc == 0 && (pl.t0 == lo.t0 || pl.t1 == lo.t0)) {
          bo.pad = pl.t0 == lo.t0 ? 1 : 2;
          group_ops[group_of[prev]].push_back(bo);
          if (!(lo.flags & LF_DIAG)) pl.flags &= ~LF_DIAG;
          p->lowered_src[prev].push_back((int)i);
          continue;
        }
      }
    }
    std::vector<BuildOp> taken;      // pending 1-q matrices a new 4x4 absorbs (they act first)
    std::vector<int> taken_src;
    if (merge_2q && lo.kind == LK_2Q && lo.nc == 0) {
      const int pa = last_touch[lo.t0], pb = last_touch[lo.t1];
      if (pa >= 0 && pa == pb) {
        LoweredOp &pl = p->lowered[pa];
        if (pl.kind == LK_2Q && pl.nc == 0 && pl.t0 == lo.t0 && pl.t1 == lo.t1) {
          group_ops[group_of[pa]].push_back(bo);
          if (!(lo.flags & LF_DIAG)) pl.flags &= ~LF_DIAG;
          p->lowered_src[pa].push_back((int)i);
          continue;
        }
      }
      for (int side = 0; side < 2; ++side) {
        const int t = side == 0 ? lo.t0 : lo.t1;
        const int prev = last_touch[t];
        if (prev < 0) continue;
        LoweredOp &pl = p->lowered[prev];
        if (!(pl.kind == LK_1Q && pl.nc == 0 && pl.t0 == t)) continue;
        for (BuildOp b1 : group_ops[group_of[prev]]) {
          b1.pad = (uint16_t)(side + 1);
          taken.push_back(b1);
        }
        if (!(pl.flags & LF_DIAG)) lo.flags &= ~LF_DIAG;
        for (int s_ : p->lowered_src[prev]) taken_src.push_back(s_);
        group_ops[group_of[prev]].clear();  // (its matrix slot stays allocated, nothing builds or reads it)
        pl.kind = kLoweredDead;             // removed below
      }
    }

    const uint32_t dim = lo.kind == LK_2Q ? 4u : 2u;
    lo.mat_off = p->mat_floats;
    p->mat_floats += dim * dim * 2;
    const int idx = (int)p->lowered.size();
    group_of.resize(idx + 1, -1);
    group_of[idx] = (int)group_ops.size();
    taken.push_back(bo);
    group_ops.push_back(taken);
    group_meta.push_back({lo.mat_off, dim});
    const uint64_t m = op_mask(lo, n);
    for (int b = 0; b < n; ++b)
      if (m & bit(b)) last_touch[b] = idx;
    p->lowered.push_back(lo);
    taken_src.push_back((int)i);
    std::sort(taken_src.begin(), taken_src.end());
    p->lowered_src.push_back(taken_src);
  }
  {  // drop the 1-q operators a 4x4 absorbed
    size_t w = 0;
    for (size_t r = 0; r < p->lowered.size(); ++r) {
      if (p->lowered[r].kind == kLoweredDead) continue;
      if (w != r) {
        p->lowered[w] = p->lowered[r];
        p->lowered_src[w] = std::move(p->lowered_src[r]);
      }
      ++w;
    }
    p->lowered.resize(w);
    p->lowered_src.resize(w);
  }
  // flatten the per-matrix source lists (tape order inside each group)
  for (size_t g = 0; g < group_ops.size(); ++g) {
    if (group_ops[g].empty()) continue;
    BuildGroup bg{(uint32_t)p->build_ops.size(), 0, group_meta[g].first, group_meta[g].second};
    for (const BuildOp &b : group_ops[g]) p->build_ops.push_back(b);
    bg.end = (uint32_t)p->build_ops.size();
    p->groups.push_back(bg);
  }

  // ---- 1b. low bit positions first ---------------------------------------------
  // Gates that share no wire commute: among the gates that are ready (no earlier gate on any
  // of their wires pending) take the one whose highest bit position is lowest.  The passes
  // then work their way up from the low bits, and a run from |0..0> keeps its known-zero
  // bits at the TOP: the amplitudes that can be non-zero stay one contiguous block (the
  // measuring pass of K2 reads 8 MiB in one piece instead of 128-byte runs every 2 KiB).
  if (fuse && !(p->flags & QMLE_PLAN_TAPE_ORDER) && p->lowered.size() > 1 && p->lowered.size() <= 16384) {
    const size_t nl = p->lowered.size();
    std::vector<std::vector<int>> queue(n);  // per bit: ops touching it, tape order
    std::vector<size_t> head(n, 0);
    std::vector<uint64_t> masks(nl);
    for (size_t i = 0; i < nl; ++i) {
      masks[i] = op_mask(p->lowered[i], n);
      for (int b = 0; b < n; ++b)
        if (masks[i] & bit(b)) queue[b].push_back((int)i);
    }
    auto ready = [&](int i) {
      for (int b = 0; b < n; ++b)
        if ((masks[i] & bit(b)) && queue[b][head[b]] != i) return false;
      return true;
    };
    std::vector<std::pair<int, int>> heap;  // (highest bit, index), min-heap
    auto cmp = [](const std::pair<int, int> &x, const std::pair<int, int> &y) { return x > y; };
    std::vector<char> queued(nl, 0);
    auto push_if_ready = [&](int i) {
      if (queued[i] || !ready(i)) return;
      queued[i] = 1;
      heap.push_back({63 - __builtin_clzll(masks[i] ? masks[i] : 1ull), i});
      std::push_heap(heap.begin(), heap.end(), cmp);
    };
    for (size_t i = 0; i < nl; ++i) push_if_ready((int)i);
    std::vector<int> order;
    order.reserve(nl);
    while (!heap.empty()) {
      std::pop_heap(heap.begin(), heap.end(), cmp);
      const int i = heap.back().second;
      heap.pop_back();
      order.push_back(i);
      for (int b = 0; b < n; ++b)
        if (masks[i] & bit(b)) ++head[b];
      for (int b = 0; b < n; ++b)
        if ((masks[i] & bit(b)) && head[b] < queue[b].size()) push_if_ready(queue[b][head[b]]);
    }
    if (order.size() == nl) {
      std::vector<LoweredOp> lo2(nl);
      std::vector<std::vector<int>> src2(nl);
      for (size_t k = 0; k < nl; ++k) {
        lo2[k] = p->lowered[order[k]];
        src2[k] = std::move(p->lowered_src[order[k]]);
      }
      p->lowered.swap(lo2);
      p->lowered_src.swap(src2);
    }
  }

  // ---- 2. choose regime ------------------------------------------------------
  const int forced_T = (int)((p->flags >> 8) & 0xff);
  const int forced_L = (int)((p->flags >> 16) & 0xff);
  p->whole_state_lds = (n <= kLdsMaxQubits) && !(p->flags & QMLE_PLAN_FORCE_GLOBAL) &&
                       (forced_T == 0 || forced_T >= n);
  if (!p->whole_state_lds && forced_T != 0 && forced_T < 4 && forced_T < n)
    return QMLE_ERR_INVALID_ARG;

  // ---- 3. schedule into stages (for one tile geometry) ---------------------------
  // lazy: X / CX whose wires are not in the tile yet wait (the fast tile path folds them into
  // the LDS layout for free, so they should never claim a tile bit a dense gate could use);
  // whatever room is left after the first sweep is filled by a second, plain greedy sweep.
  // HE layer at n = 24: dense gates per pass 12/7/5 -> 12/8/4, the measuring pass drops from
  // two register-tile groups to one.
  // carry >= 0: every tile after the first also holds bit position `carry` (position 6 = byte
  // address bit 9: a read+write pass whose tile holds it moves two 128-byte rows 512 B apart with
  // every load / store instruction and runs 12-15 % faster -- 51 -> 44 us per state at n = 24,
  // profiles/r03_rw_tile_bits.txt; the price is one of the tile's 8 free positions).
  // T_first > T: the first stage of a run from |0..0> computes ONE tile per state (everything else
  // is zeros whatever the tile size), so it may stage up to 2^14 amplitudes at no cost in traffic.
  // top_first (round 5, tuner candidates only): the first stage of a run from |0..0> -- ONE tile per state wherever
  // that tile sits -- takes the TOP T_first positions instead of the low ones: every gate that lives inside that
  // window runs on one tile per state, and what is left (the low positions plus whatever a deferred gate reaches
  // up to) is scheduled as usual, on tiles with long contiguous rows.  The n = 24 headline layer: {10..23} first,
  // then ONE measuring pass on {0..10, 23} -- two passes where the low-first schedule needs three.  The tile is a
  // contiguous run of positions (Stage::shift): local index << shift is the address, no rows, no LUT.
  auto schedule = [&](int T, int L, bool lazy = false, int carry = -1, int T_first = 0, bool top_first = false) {
    if (p->whole_state_lds) T = n;
    if (T > kLdsMaxQubits) T = kLdsMaxQubits;
    if (T > n) T = n;
    if (L > T) L = T;
    if (L < 1) L = 1;
    p->tile_T = T;
    p->tile_L = L;
    p->stages.clear();
    p->dev_ops.clear();
    p->dev_src.clear();
    p->op_groups.clear();
    p->ops2.clear();
    p->groups2.clear();
    p->tbl2.clear();
    p->consts.resize(n_user_consts);  // drop permuted-matrix copies of a previous candidate
    const size_t nl = p->lowered.size();
    std::vector<char> done(nl, 0);
    size_t n_done = 0;
    const bool no_fusion = (p->flags & QMLE_PLAN_NO_FUSION) != 0;
    const bool force_tile = (p->flags & QMLE_PLAN_FORCE_TILE) != 0;
    const uint64_t all_mask = n >= 64 ? ~0ull : bit(n) - 1;
    uint32_t Zrun = n >= 32 ? ~0u : ((1u << n) - 1u);  // known-zero positions so far (|0..0> start)

    auto popc = [](uint64_t x) { return __builtin_popcountll(x); };

    const int T_plan = T;
    while (n_done < nl) {
      std::vector<int> members;
      uint64_t Q = 0;
      int stageL = L;
      T = T_plan;
      if (!p->whole_state_lds && p->stages.empty() && T_first > T_plan) {
        T = T_first;
        if (T > kLdsMaxQubits) T = kLdsMaxQubits;
        if (T > n - 1) T = n - 1;
        if (T < T_plan) T = T_plan;
      }
      bool top_stage = false;
      if (top_first && !p->whole_state_lds && p->stages.empty() && T < n && n <= 28 && !no_fusion && !force_tile) {
        const uint64_t W = all_mask & ~(bit(n - T) - 1);
        uint64_t blocked = 0;
        for (size_t i = 0; i < nl; ++i) {
          const LoweredOp &o = p->lowered[i];
          const uint64_t m = op_mask(o, n);
          if (o.kind == LK_DIAG_ALL || (m & blocked) || (m & ~W)) blocked |= m;
          else members.push_back((int)i);
          if ((blocked & all_mask) == all_mask) break;
        }
        if (!members.empty() && members.size() < nl) {
          top_stage = true;
          Q = W;
        } else {
          members.clear();  // nothing lives up there (or everything does: one stage, nothing to gain)
        }
      }
      if (top_stage) {
        // (members chosen above)
      } else if (p->whole_state_lds) {
        for (size_t i = 0; i < nl; ++i) members.push_back((int)i);
        Q = all_mask;
      } else {
        // first pending op decides whether the default low-bit count fits
        size_t first = 0;
        while (done[first]) ++first;
        const LoweredOp &fo = p->lowered[first];
        if (fo.kind == LK_DIAG_ALL) {
          Stage st;
          st.kind = ST_DIAG_ALL;
          st.op_begin = (int)p->dev_ops.size();
          p->dev_ops.push_back(fo);
          p->dev_src.push_back(p->lowered_src[first].size() == 1 ? p->lowered_src[first][0] : -1);
          st.op_end = (int)p->dev_ops.size();
          st.src_ops = p->lowered_src[first];
          p->stages.push_back(st);
          done[first] = 1;
          ++n_done;
          continue;
        }
        const uint64_t fm = op_mask(fo, n);
        while (stageL > 1 && popc((bit(stageL) - 1) | fm) > T) --stageL;
        Q = bit(stageL) - 1;
        if (carry >= stageL && carry < n && !p->stages.empty() && popc(Q | fm | bit(carry)) <= T) Q |= bit(carry);
        std::vector<char> taken;
        if (lazy) taken.assign(nl, 0);
        for (int sweep = lazy ? 0 : 1; sweep < 2; ++sweep) {
          uint64_t blocked = 0;
          for (size_t i = first; i < nl; ++i) {
            if (done[i] || (lazy && taken[i])) continue;
            const LoweredOp &o = p->lowered[i];
            const uint64_t m = op_mask(o, n);
            const bool wait = sweep == 0 && o.kind == LK_1Q && o.nc <= 1 && (o.flags & LF_PERMX) &&
                              (m & ~Q) != 0;
            if (o.kind == LK_DIAG_ALL || (m & blocked)) {
              blocked |= m;
            } else if (!wait && popc(Q | m) <= T) {
              Q |= m;
              members.push_back((int)i);
              if (lazy) taken[i] = 1;
              if (no_fusion) break;
            } else {
              blocked |= m;
            }
            if ((blocked & all_mask) == all_mask) break;
          }
          if (popc(Q) >= T) break;
        }
        // (an op taken by the second sweep never shares a wire with a LATER op of the first: that
        // one would have been blocked behind it -- so index order is a valid execution order)
        if (lazy) std::sort(members.begin(), members.end());
      }

      Stage st;
      st.L = stageL;
      st.op_begin = (int)p->dev_ops.size();
      const LoweredOp &m0 = p->lowered[members[0]];
      // a control on bits 1..3 selects 16/32/64-byte runs inside every 128-byte line: the
      // streaming kernel would move whole lines for half the work (measured 2-4x slower
      // than a tile pass at n = 28), so those go through the LDS tile instead
      // (round 2: with the target on bits 1..6 too, the lane-exchange mode of the streaming
      // kernel takes those controls: k_direct_1q mode 7)
      const bool direct_ok = !top_stage && !p->whole_state_lds && !force_tile && members.size() == 1 &&
                             m0.kind == LK_1Q &&
                             (m0.nc == 0 ||
                              (m0.nc == 1 && (m0.c0 == 0 || m0.c0 >= 4 ||
                                              (!(m0.flags & LF_DIAG) && n >= 14 && m0.t0 >= 1 && m0.t0 <= 6) ||
                                              // a controlled PHASE rewrites the |11> quarter only (k_direct_1q mode 9):
                                              // with the target above the line that is half the lines, not all of them
                                              ((m0.flags & LF_PHASE) && m0.t0 >= 4))));
      if (direct_ok) {
        st.kind = ST_DIRECT;
        p->dev_ops.push_back(m0);
        p->dev_src.push_back(p->lowered_src[members[0]].size() == 1 ? p->lowered_src[members[0]][0] : -1);
      } else {
        st.kind = ST_TILE;
        // pad the tile with the lowest free bit positions
        // (tuning: QMLE_PAD_HIGH=1 pads the LAST stage from the top instead)
        const int pad_high_env = p->pad_high >= 0 ? p->pad_high
                                 : std::getenv("QMLE_PAD_HIGH") ? atoi(std::getenv("QMLE_PAD_HIGH")) : 0;
        if (pad_high_env && !p->stages.empty() && n_done + members.size() == nl) {
          if (carry >= stageL && carry < n && (Q & bit(carry))) {  // the carried position serves read+write passes
            uint64_t need = 0;
            for (int mi : members) need |= op_mask(p->lowered[mi], n);
            if (!(need & bit(carry))) Q &= ~bit(carry);
          }
          if (pad_high_env >= 2 && popc(Q) < T) Q |= bit(pad_high_env);  // (one chosen low position)
          for (int b = n - 1; b >= 0 && popc(Q) < T; --b) Q |= bit(b);
        }
        // The LAST stage of a 2^13-tile schedule whose gates need <= 12 positions takes a 2^12 tile: five 32 KiB
        // workgroups per CU overlap their loads and their groups where two 64 KiB ones do not (the measuring pass
        // of the 4-layer n = 24 model: experiment QMLE_SMALL_LAST_TILE)
        int Tpad = T;
        {
          const char *e = std::getenv("QMLE_SMALL_LAST_TILE");
          const bool on = e ? atoi(e) != 0 : kSmallLastTileDefault;
          if (on && T == 13 && !p->stages.empty() && n_done + members.size() == nl && popc(Q) <= 12 && !no_fusion) Tpad = 12;
        }
        for (int b = 0; b < n && popc(Q) < Tpad; ++b) Q |= bit(b);
        st.T = popc(Q);
        int nt = 0, no = 0;
        int8_t local_of[64];
        for (int b = 0; b < n; ++b) {
          if (Q & bit(b)) { local_of[b] = (int8_t)nt; st.tile_bits[nt++] = (int8_t)b; }
          else { local_of[b] = -1; st.outer_bits[no++] = (int8_t)b; }
        }
        // contiguous low run actually present
        int run = 0;
        while (run < st.T && st.tile_bits[run] == run) ++run;
        st.L = run < 1 ? 1 : run;
        if (top_stage) {  // a contiguous run of positions that does not start at 0: address = local index << shift
          st.shift = n - st.T;
          st.L = st.T;
        }
        for (int mi : members) {
          LoweredOp o = p->lowered[mi];
          if (o.kind != LK_DIAG_ALL) {
            o.t0 = local_of[(int)o.t0];
            if (o.t1 >= 0) o.t1 = local_of[(int)o.t1];
            if (o.c0 >= 0) o.c0 = local_of[(int)o.c0];
            if (o.c1 >= 0) o.c1 = local_of[(int)o.c1];
          }
          p->dev_ops.push_back(o);
          p->dev_src.push_back(p->lowered_src[mi].size() == 1 ? p->lowered_src[mi][0] : -1);
        }
      }
      st.op_end = (int)p->dev_ops.size();
      st.n_tile_ops = st.op_end - st.op_begin;
      if (st.kind == ST_TILE) {
        const std::vector<LoweredOp> tile_local(p->dev_ops.begin() + st.op_begin,
                                                p->dev_ops.begin() + st.op_end);
        uint32_t zin_local = 0;  // known-zero bits when this stage starts, tile-local
        for (int j = 0; j < st.T; ++j)
          if (Zrun & (1u << st.tile_bits[j])) zin_local |= 1u << j;
        build_fast_groups(p, st, tile_local, zin_local);
        group_stage_ops(p, st);
      }
      for (int mi : members) {
        done[mi] = 1;
        ++n_done;
        const LoweredOp &lo = p->lowered[mi];  // global positions
        if (lo.kind == LK_4Q) {
          st.touched |= (1u << lo.t0) | (1u << lo.t1) | (1u << lo.c0) | (1u << lo.c1);
        } else if (lo.kind != LK_DIAG_ALL && !(lo.flags & LF_DIAG)) {
          st.touched |= 1u << lo.t0;  // controls keep their known-zero status
          if (lo.t1 >= 0) st.touched |= 1u << lo.t1;
        }
        for (int s : p->lowered_src[mi]) {
          st.src_ops.push_back(s);
          st.algo_bytes_per_state += algo_bytes(p->ops[s], n);
        }
      }
      Zrun &= ~st.touched;
      p->stages.push_back(st);
    }
    if (p->whole_state_lds && p->stages.empty()) {
      // empty circuit: still need one stage to produce |0...0>
      Stage st;
      st.kind = ST_TILE;
      st.T = n;
      st.L = L;
      for (int b = 0; b < n; ++b) st.tile_bits[b] = (int8_t)b;
      p->stages.push_back(st);
    }
    // known-zero bit positions along the execution order (|0..0> start)
    uint32_t Z = n >= 32 ? ~0u : ((1u << n) - 1u);
    for (size_t si = 0; si < p->stages.size(); ++si) {
      Stage &st = p->stages[si];
      st.zero_in = Z;
      Z &= ~st.touched;
      st.next_tile = si + 1 < p->stages.size() && p->stages[si + 1].kind == ST_TILE;
      st.product_ok = false;
      if (st.kind == ST_TILE && si > 0 && st.grp_end > st.grp_begin && st.T >= 8) {
        uint32_t seen = 0;
        bool ok = true;
        for (int g = st.grp_begin; g < st.grp_end && ok; ++g) {
          const OpGroup &og = p->op_groups[g];
          ok = og.kind == GK_REG4;
          for (int j = 0; j < 4 && ok; ++j) {
            const uint32_t lb = 1u << og.bits[j];
            ok = !(seen & lb) && ((st.zero_in >> st.tile_bits[og.bits[j]]) & 1u);
            seen |= lb;
          }
        }
        st.product_ok = ok;
      }
    }
    p->fold_groups = 0;
    for (const Stage &st : p->stages)
      if (st.product_ok && st.grp_end - st.grp_begin > p->fold_groups)
        p->fold_groups = st.grp_end - st.grp_begin;

  };

  // pass-cost model (microseconds per state at n = 24; measured on MI355X, dense passes of 1-
  // and 3-layer HE circuits, tools/stage_profile.py): a tile pass costs ~11 for its HBM round
  // trip and ~25 per register-tile group (the gate arithmetic is what a pass is made of: 62 us
  // with 2 groups, 125 with 5, 192 with 7); a direct single-gate pass ~33
  // Known zeros scale both parts: a stage reads 2^-|zero_in| of the state, computes and (when
  // the next stage is a tile stage) stores only the tiles whose outer bits are live.
  const bool sparse_model = !(p->flags & (QMLE_PLAN_NO_SPARSE | QMLE_PLAN_PREFETCH));
  // the plan is only ever run from |0..0> (qmle_run_batch): set on the variants / children that
  // qmle_plan_create compiles for that purpose, never on a plan handed to qmle_apply_inplace
  const bool zero_run = (p->flags & QMLE_PLAN_INTERNAL_ZERO_RUN) != 0;
  auto cost = [&]() {
    double c = 0;
    for (size_t si = 0; si < p->stages.size(); ++si) {
      const Stage &st = p->stages[si];
      if (st.kind != ST_TILE) { c += 33.0; continue; }
      double rd = si == 0 ? 0.0 : 1.0, wr = 1.0, tiles = 1.0;
      if (sparse_model && st.zero_in) {
        uint32_t outer = 0;
        for (int i = 0; i < n - st.T; ++i) outer |= 1u << st.outer_bits[i];
        tiles = std::ldexp(1.0, -__builtin_popcount(st.zero_in & outer));
        if (si > 0) rd = std::ldexp(1.0, -__builtin_popcount(st.zero_in));
        if (st.next_tile) wr = tiles;
      }
      // the first stage of a run from |0..0> runs its gates on one tile per state (the rest is a fill)
      if (si == 0 && zero_run && !sparse_model) tiles = std::ldexp(1.0, st.T - n);
      const bool last = si + 1 == p->stages.size();
      if (st.fast_ok) {
        // Fast kernel (k_tile2), round-3 model, fitted to per-pass HIP-event times of 1- and 4-layer
        // circuits at n = 24 with and without their gate groups (tools/deep_anatomy.py,
        // QMLE_DBG_T2=1; profiles/r03_pass_model.txt): a pass takes the LONGER of its memory time
        // and its compute time plus a sixth of the shorter one.
        //   memory: 19.5 us write-only, 21 read-only, 50 read+write (5.4 TB/s: the mix costs);
        //   a read+write pass whose every load / store instruction spans positions 6 and 13 (byte
        //   address bits 9 and 16) and no other position below 8 -- a wave's 8 rows of 128 B are
        //   the tile's three lowest high positions -- runs 15 % faster (51.3 -> 43.3 us, K2 pass
        //   2; 54 -> 45 in the bare pass for every tile {6, 13, x >= 8, ...} and for no other pair
        //   tried); every pair of high positions 8 apart costs ~4.5 us (54 -> 71 for {12-15,
        //   20-23}).  The HBM channel hash behind both is not documented: modelled as measured.
        //   compute: 5 + 9.5 per register-tile group, + 10 for the <Z> sums of a measuring pass.
        double rw = (si == 0 ? 0.0 : 21.0 * rd) + (last ? 0.0 : 19.5 * wr);
        if (si > 0 && !last) {
          rw += 9.5 * (rd < wr ? rd : wr);
          double shape = 0.0;
          if (st.L == 4 && st.T >= 12) {
            const int h0 = st.tile_bits[4], h1 = st.tile_bits[5], h2 = st.tile_bits[6];
            if (h0 == 6 && h1 >= 8 && (h1 == 13 || h2 == 13)) shape -= 7.5;
          }
          uint32_t hi = 0;
          for (int j = st.L; j < st.T; ++j) hi |= 1u << st.tile_bits[j];
          shape += 4.5 * __builtin_popcount(hi & (hi >> 8));
          rw += shape * (rd < wr ? rd : wr);
        }
        const double cmp = (5.0 + 9.5 * (st.fast_end - st.fast_begin) + (last ? 10.0 : 0.0)) * tiles;
        c += 1.5 + 0.8 * std::ldexp(1.0, 24 - n) + (rw > cmp ? rw + cmp / 6.0 : cmp + rw / 6.0);
        continue;
      }
      const double mem = rd + (last ? 0.0 : wr);
      c += 2.0 + 0.8 * std::ldexp(1.0, 24 - n) + 6.0 * mem +
           25.0 * (st.grp_end - st.grp_begin) * tiles;
    }
    // a schedule that ends in a streaming pass leaves the measurement to a pass of its own (one
    // more read of the state), where a tile pass measures on the fly
    if (!p->stages.empty() && p->stages.back().kind != ST_TILE) c += 21.0;
    return c;
  };
  if (p->whole_state_lds || forced_T != 0 || forced_L != 0 || no_fusion_flag) {
    schedule(forced_T ? forced_T : kDefaultTileBits, forced_L ? forced_L : kDefaultLowBits);
  } else {
    static const int cand[][2] = {{13, 7}, {13, 5}, {12, 4}, {13, 4}, {12, 5}, {13, 6}};
    // k = geometry (6) x [eager, lazy CX] (2) x variant (4): 0 plain, 1 wide first stage, 2 position 6
    // carried + wide first stage, 3 position 6 carried.  Ties keep the lower index (round-2 schedules).
    int best = 0;
    double best_cost = 1e300;
    static const bool no_lazy = std::getenv("QMLE_NO_LAZY_CX") != nullptr;
    static const bool no_carry = std::getenv("QMLE_NO_CARRY6") != nullptr;
    static const bool no_wide = std::getenv("QMLE_NO_WIDE_FIRST") != nullptr;
    // (round 5) variant 4, k in [48, 60): first tile of 2^14 amplitudes on the TOP positions (all-live runs from
    // |0..0>).  Such a schedule trades an HBM pass for arithmetic in the passes that are left: fewer bytes, less time, a
    // lower fraction of the HBM roofline.  QMLE_NO_TOP_FIRST=1: the round-4 candidates only.
    const bool no_top = std::getenv("QMLE_NO_TOP_FIRST") != nullptr || std::getenv("QMLE_NO_INIT_FILL") != nullptr;  // (read per compile: bench.py's k2_three_pass leg)
    auto run_cand = [&](int k) {
      const int g = k % 6, lazy = (k / 6) % 2, v = k / 12;
      const bool wide = v == 1 || v == 2 || v == 4, carry6 = v == 2 || v == 3;
      schedule(cand[g][0], cand[g][1], lazy != 0, carry6 ? 6 : -1, wide ? kLdsMaxQubits : 0, v == 4);
    };
    auto allowed = [&](int k) {
      const int g = k % 6, lazy = (k / 6) % 2, v = k / 12;
      if (cand[g][0] >= n) return false;
      if (lazy && no_lazy) return false;
      // (known-zero runs keep the round-2 schedules: their first passes are launch-bound special
      // kernels tuned for the (12, 4) geometry, and the wide first tile cost them 4-8 % at n = 24)
      if (v >= 1 && sparse_model && std::getenv("QMLE_SPARSE_VARIANTS") == nullptr) return false;
      if ((v == 1 || v == 2) && (!zero_run || no_wide)) return false;
      if ((v == 2 || v == 3) && (no_carry || cand[g][1] != 4 || n < 16)) return false;
      if (v == 4 && (!zero_run || sparse_model || no_top || n < 16 || n > 28 || (p->flags & QMLE_PLAN_PREFETCH))) return false;
      return true;
    };
    p->cand_ranking.clear();
    for (int k = 0; k < 60; ++k) {
      if (!allowed(k)) continue;
      run_cand(k);
      if (k >= 48 && (p->stages.empty() || p->stages[0].shift == 0)) continue;  // (no top-first stage came of it)
      const double c = cost();
      p->cand_ranking.push_back({c, k});
      // (the round-3 variants must win by 2 %: the model knows their effect from two circuits; a top-first schedule
      // by 5 % -- it wins by dropping a whole pass or not at all: K2 headline 58.2 against 92.5 predicted, 60.5 against
      // 88.5 ms measured)
      const double margin = k >= 48 ? (best < 48 ? 0.95 : 1.0) : (k >= 12 && best < 12 ? 0.98 : 1.0);
      if (c < best_cost * margin - 1e-9) { best_cost = c; best = k; }
    }
    // (tuning only: force one of the candidates to measure it against the model's choice)
    std::sort(p->cand_ranking.begin(), p->cand_ranking.end());
    const int force = p->force_candidate >= 0 ? p->force_candidate
                      : std::getenv("QMLE_FORCE_CAND") ? atoi(std::getenv("QMLE_FORCE_CAND")) : -1;  // (read per compile: tools/cand_sweep.py)
    if (force >= 0 && force < 60 && (force < 48 || allowed(force)) && cand[force % 6][0] < n && (force < 12 || zero_run)) best = force;
    run_cand(best);
    p->chosen_candidate = best;
  }
  p->model_cost = cost();
  // ---- matrices no forward kernel reads --------------------------------------------------------
  // An X / CX inside a register-tile group is a swap of amplitudes (reg_dispatch<2>, f_x / f_cx) or a
  // change of the LDS layout map (build_fast_groups); its 2x2 matrix is never read by a tile pass.  The
  // per-sample matrix builder spent 46 % of its groups on them in the Fourier-grid model (60 CX of 130
  // groups, a fifth of a saturated 10-qubit batch with the rest of the builder).  The
  // build groups are ordered needed-first; the forward engine builds [0, n_groups_needed), the adjoint
  // sweep and the complex128 engine (which apply a CX through its matrix) all of them.
  {
    std::vector<char> unread(p->mat_floats + 1, 0);
    for (const Stage &st : p->stages) {
      if (st.kind != ST_TILE || st.T < 4 || (p->flags & QMLE_PLAN_NO_REGTILE)) continue;  // (group_stage_ops: regs_ok)
      for (int i = st.op_begin; i < st.op_end; ++i) {
        const LoweredOp &o = p->dev_ops[i];
        if (o.kind == LK_1Q && o.nc <= 1 && (o.flags & LF_PERMX)) unread[o.mat_off] = 1;
      }
    }
    std::stable_partition(p->groups.begin(), p->groups.end(),
                          [&](const BuildGroup &g) { return !(g.dim == 2 && unread[g.mat_off]); });
    p->n_groups_needed = 0;
    for (const BuildGroup &g : p->groups)
      if (!(g.dim == 2 && unread[g.mat_off])) ++p->n_groups_needed;
    static const bool build_all = std::getenv("QMLE_BUILD_ALL_MATRICES") != nullptr;
    if (build_all) p->n_groups_needed = (int)p->groups.size();
  }
  return QMLE_OK;
}

// HBM bytes per state a stage moves in a run from |0..0> (TM_STORE epilogue; a fused
// measurement in the last stage writes nothing): known-zero amplitudes are not read, tiles of
// zeros are not stored when the next stage is a tile stage (Stage::zero_in / next_tile).
static double stage_read_bytes(const qmle_plan *p, size_t si) {
  const Stage &st = p->stages[si];
  const double D = std::ldexp(1.0, p->n);
  const bool sparse = !(p->flags & (QMLE_PLAN_NO_SPARSE | QMLE_PLAN_PREFETCH));
  if (st.kind != ST_TILE) return st.kind == ST_DIRECT ? 0.5 * st.algo_bytes_per_state : 8.0 * D;
  if (si == 0) return 0.0;  // generated in LDS
  if (!sparse) return 8.0 * D;
  uint32_t z = st.zero_in;
  int nz = __builtin_popcount(z & ~1u);  // 16-byte loads: bit 0 rides along
  return 8.0 * std::ldexp(1.0, p->n - nz);
}
static double stage_write_bytes(const qmle_plan *p, size_t si) {
  const Stage &st = p->stages[si];
  const double D = std::ldexp(1.0, p->n);
  const bool sparse = !(p->flags & (QMLE_PLAN_NO_SPARSE | QMLE_PLAN_PREFETCH));
  if (st.kind != ST_TILE) return st.kind == ST_DIRECT ? 0.5 * st.algo_bytes_per_state : 8.0 * D;
  if (!sparse || !st.next_tile) return 8.0 * D;
  uint32_t outer = 0;
  for (int i = 0; i < p->n - st.T; ++i) outer |= 1u << st.outer_bits[i];
  return 8.0 * std::ldexp(1.0, p->n - __builtin_popcount(st.zero_in & outer));
}

int expval_kernel_of(const qmle_plan *p, size_t si, bool sparse) {
  const Stage &st = p->stages[si];
  if (st.kind != ST_TILE || si == 0 || (p->flags & QMLE_PLAN_PREFETCH)) return 0;
  if (st.grp_end - st.grp_begin != 1 || p->op_groups[st.grp_begin].kind != GK_REG4) return 0;
  if (st.T < 10 || st.T > 14 || st.T >= p->n || (st.op_end - st.op_begin) > 1000) return 0;
  const OpGroup &g = p->op_groups[st.grp_begin];
  int live_bits = 0;  // register bits that are not known-zero on input
  for (int j = 0; j < 4; ++j)
    live_bits += !(sparse && ((st.zero_in >> st.tile_bits[g.bits[j]]) & 1u));
  if (live_bits == 0 && p->n - st.T >= 5) return 3;
  return live_bits <= 2 && g.n_ops > 0 ? 2 : 1;
}

// fp32 flops per state of the operators the plan really applies (after 1-qubit merging): a dense
// 2x2 costs 4 complex multiplies + 2 complex adds per amplitude pair = 14 per amplitude, a
// diagonal one 6, a permutation 0; a control halves the amplitudes touched.
static double plan_flops_per_state(const qmle_plan *p) {
  const double D = std::ldexp(1.0, p->n);
  double f = 0;
  for (const LoweredOp &op : p->lowered) {
    const double live = D / (double)(1u << op.nc);
    switch (op.kind) {
      case LK_1Q: f += (op.flags & LF_PERMX) ? 0.0 : (op.flags & LF_DIAG) ? 6.0 * live : 14.0 * live; break;
      case LK_2Q: f += 30.0 * live; break;
      case LK_DIAG_ALL: f += 6.0 * D; break;
      case LK_4Q: f += 126.0 * D; break;
    }
  }
  return f;
}

// the same count for the operators of one stage.  `live`: in a run from |0..0> with known-zero tracking, an
// operator is charged for the amplitudes that can be non-zero when it is applied (every known-zero position
// outside its own bits halves them; a known-zero CONTROL leaves nothing to do; a gate that is not diagonal takes its
// target out of the set) -- what the kernels skip at wave granularity; else nominal, the whole register.
static double stage_flops_per_state(const qmle_plan *p, const Stage &st, bool live_only) {
  // (the first stage of an all-live run from |0..0> computes ONE tile per state -- the rest is a fill)
  const bool one_tile = !p->stages.empty() && &st == &p->stages[0] && st.kind == ST_TILE && st.T < p->n &&
                        (p->flags & QMLE_PLAN_INTERNAL_ZERO_RUN) && (p->flags & QMLE_PLAN_NO_SPARSE);
  const double D = std::ldexp(1.0, one_tile ? st.T : p->n);
  double f = 0;
  const bool sparse = live_only && !(p->flags & (QMLE_PLAN_NO_SPARSE | QMLE_PLAN_PREFETCH));
  uint32_t Z = sparse ? st.zero_in : 0u;
  // bit indices of a tile stage's ops: GROUP-local (0..3) inside a register-tile group, else tile-local
  std::vector<const OpGroup *> group_of(st.op_end > st.op_begin ? st.op_end - st.op_begin : 0, nullptr);
  if (st.kind == ST_TILE)
    for (int g = st.grp_begin; g < st.grp_end; ++g) {
      const OpGroup &og = p->op_groups[g];
      if (og.kind != GK_REG4 && og.kind != GK_REG4X) continue;
      for (int k = 0; k < (int)og.n_ops; ++k) {
        const int idx = (int)og.op_begin + k - st.op_begin;
        if (idx >= 0 && idx < (int)group_of.size()) group_of[idx] = &og;
      }
    }
  const OpGroup *cur = nullptr;
  auto gpos = [&](int b) {
    if (st.kind != ST_TILE || b < 0) return b;
    const int local = cur ? (int)cur->bits[b & 3] : b;
    return (int)st.tile_bits[local];
  };
  for (int i = st.op_begin; i < st.op_end && i < (int)p->dev_ops.size(); ++i) {
    const LoweredOp &op = p->dev_ops[i];
    cur = group_of.empty() ? nullptr : group_of[i - st.op_begin];
    uint32_t own = 0, ctl = 0;
    if (op.kind != LK_DIAG_ALL) {
      own |= 1u << gpos(op.t0);
      if (op.t1 >= 0) own |= 1u << gpos(op.t1);
      if (op.kind == LK_4Q) { own |= 1u << gpos(op.c0); own |= 1u << gpos(op.c1); }
      else {
        if (op.nc >= 1) ctl |= 1u << gpos(op.c0);
        if (op.nc >= 2) ctl |= 1u << gpos(op.c1);
      }
    }
    double amps = D / (double)(1u << op.nc);
    if (Z & ctl) amps = 0.0;
    else amps /= (double)(1ull << __builtin_popcount(Z & ~own & ~ctl));
    switch (op.kind) {
      case LK_1Q: f += (op.flags & LF_PERMX) ? 0.0 : (op.flags & LF_DIAG) ? 6.0 * amps : 14.0 * amps; break;
      case LK_2Q: f += 30.0 * amps; break;
      case LK_DIAG_ALL: f += 6.0 * amps; break;
      case LK_4Q: f += 126.0 * amps; break;
    }
    if (op.kind == LK_DIAG_ALL || (op.kind == LK_1Q && (op.flags & LF_DIAG))) continue;
    Z &= ~own;  // mixed positions are live from here on
  }
  return f;
}

std::string describe_plan(const qmle_plan *p) {
  std::ostringstream os;
  os << "{\"n_qubits\":" << p->n << ",\"n_ops\":" << p->ops.size()
     << ",\"n_lowered\":" << p->lowered.size()
     << ",\"whole_state_lds\":" << (p->whole_state_lds ? "true" : "false")
     << ",\"model_cost\":" << p->model_cost << ",\"candidate\":" << p->chosen_candidate
     << ",\"autotuned\":" << (p->autotuned ? "true" : "false")
     << ",\"zero_run\":" << ((p->flags & QMLE_PLAN_INTERNAL_ZERO_RUN) ? "true" : "false") << ",\"tile_bits\":" << p->tile_T << ",\"low_bits\":" << p->tile_L
     << ",\"mat_floats\":" << p->mat_floats
     << ",\"build_groups\":" << p->groups.size() << ",\"build_groups_needed\":" << p->n_groups_needed
     << ",\"algo_bytes_per_state\":" << p->algo_bytes_per_state
     << ",\"flops_per_state\":" << plan_flops_per_state(p) << ",\"stages\":[";
  for (size_t s = 0; s < p->stages.size(); ++s) {
    const Stage &st = p->stages[s];
    if (s) os << ",";
    os << "{\"kind\":\""
       << (st.kind == ST_DIRECT ? "direct" : st.kind == ST_TILE ? "tile" : "diag_all")
       << "\",\"n_lowered\":" << (st.op_end - st.op_begin) << ",\"T\":" << st.T
       << ",\"L\":" << st.L << ",\"shift\":" << st.shift << ",\"lds_round_trips\":" << (st.grp_end - st.grp_begin)
       << ",\"algo_bytes_per_state\":"
       << st.algo_bytes_per_state + (s + 1 == p->stages.size() ? p->extra_algo_last_stage : 0.0)
       << ",\"flops_per_state\":" << stage_flops_per_state(p, st, false)
       << ",\"flops_live_per_state\":" << stage_flops_per_state(p, st, true)
       << ",\"zero_in\":" << st.zero_in << ",\"next_tile\":" << (st.next_tile ? "true" : "false")
       << ",\"product\":" << (st.product_ok ? "true" : "false") << ",\"expval_kernel\":\""
       << (const char *[]){"k_tile", "k_reg_measure", "k_reg_measure_fold", "k_reg_measure_mono"}
              [expval_kernel_of(p, s, !(p->flags & (QMLE_PLAN_NO_SPARSE | QMLE_PLAN_PREFETCH)))]
       << "\",\"read_bytes_from_zero\":" << (unsigned long long)stage_read_bytes(p, s)
       << ",\"write_bytes_from_zero\":" << (unsigned long long)stage_write_bytes(p, s)
       << ",\"bits\":[";
    for (int i = 0; i < st.T; ++i) os << (i ? "," : "") << (int)st.tile_bits[i];
    os << "],\"groups\":[";
    for (int g = st.grp_begin; g < st.grp_end; ++g) {
      const OpGroup &og = p->op_groups[g];
      os << (g > st.grp_begin ? "," : "") << "{\"kind\":" << (int)og.kind << ",\"n_ops\":"
         << (int)og.n_ops << ",\"bits\":[" << (int)og.bits[0] << "," << (int)og.bits[1] << ","
         << (int)og.bits[2] << "," << (int)og.bits[3] << "]}";
    }
    os << "],\"fast\":" << (st.fast_ok ? "true" : "false") << ",\"fast_groups\":[";
    for (int g = st.fast_begin; g < st.fast_end; ++g)
      os << (g > st.fast_begin ? "," : "") << "{\"n_ops\":" << p->groups2[g].n_ops
         << ",\"relayout\":" << (int)p->groups2[g].relayout << "}";
    os << "],\"src_ops\":[";
    for (size_t i = 0; i < st.src_ops.size(); ++i) os << (i ? "," : "") << st.src_ops[i];
    os << "]}";
  }
  os << "]";
  if (p->expval_child)
    os << ",\"absorbed_ops\":" << p->absorbed.size()
       << ",\"expval_plan\":" << describe_plan(p->expval_child);
  os << "}";
  return os.str();
}

}  // namespace qmle
