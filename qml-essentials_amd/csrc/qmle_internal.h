// Internal structures shared by the host-side plan compiler (qmle_plan.cpp) and
// the gfx950 kernels / C-ABI entry points (the qmle_*.hip units, see qmle_host.h).  Not part of the ABI.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "qmle_sv.h"

namespace qmle {

// All kernels work in "bit position" space: wire w of an n-qubit register is bit
// p = n-1-w of the flat amplitude index (wire 0 = MSB, simulation.py:100-104).

enum LKind : uint8_t {
  LK_1Q = 0,       // (0..2 controls) x dense/diagonal 2x2 on t0
  LK_2Q = 1,       // (0..1 controls) x dense 4x4 on (t0,t1); row = 2*bit[t0] + bit[t1]
  LK_DIAG_ALL = 2, // full-register diagonal exp(-i * mark[i] * x)
  LK_4Q = 3        // dense 16x16 on (t0, t1, c0, c1) = 4 TARGET bits in wire order (MSB first);
                   // batch-constant matrix in the const blob (2-qubit Kraus superoperators)
};
enum LFlag : uint8_t {
  LF_DIAG = 1,     // matrix is diagonal
  LF_PERMX = 2,    // matrix is exactly Pauli-X (CX / CCX / X): a pure swap of amplitudes
  LF_PHASE = 4,    // diagonal with m00 == 1 exactly (CZ, ControlledPhaseShift): only target = 1 changes
};

// A run of ops applied in ONE LDS round trip: every thread gathers the 2^4 amplitudes
// spanned by `bits` into registers, applies all ops of the group there, scatters back.
enum GKind : uint8_t {
  GK_SWEEP = 0,   // one op, LDS sweep
  GK_REG4 = 1,    // run of (controlled) 2x2 ops on <= 4 bits, in registers
  GK_DENSE4 = 2,  // one LK_4Q op: gather over its 4 bits, 16x16 matrix-vector product
  GK_REG4X = 3    // a GK_REG4 run that also holds uncontrolled LK_2Q ops (dense 4x4 on two of the group's bits:
                  // the 1-wire Kraus superoperators of vec(rho), RXX ...); t0 / t1 are GROUP-local too.  Only the
                  // DENSE4 instantiations of the tile kernels run it (the common ones keep their register budget)
};
struct OpGroup {
  uint8_t kind;
  uint8_t n_ops;
  uint8_t bits[4];    // GK_REG4: tile-local bit positions, ascending
  uint8_t pad[2];
  uint32_t op_begin;  // first op (index into the plan's dev_ops); ops of a GK_REG4
                      // group carry GROUP-local bit indices 0..3 in t0 / c0
  uint32_t pad2;
};
static_assert(sizeof(OpGroup) == 16, "OpGroup layout");

// Device-visible lowered operation (16 bytes).
struct LoweredOp {
  uint8_t kind;
  uint8_t flags;
  int8_t t0, t1;     // target bit positions (t1 = -1 for LK_1Q)
  int8_t c0, c1;     // control bit positions or -1
  uint8_t nc;        // number of controls
  uint8_t pad;
  uint32_t mat_off;  // LK_1Q/LK_2Q: float offset in the per-sample matrix row
                     // LK_DIAG_ALL: float offset of the marks in the const blob
  int32_t slot;      // LK_DIAG_ALL: angle-table column, else -1
};
static_assert(sizeof(LoweredOp) == 16, "LoweredOp layout");

// One source gate feeding the per-sample matrix builder (24 bytes).
struct BuildOp {
  uint16_t opcode;
  uint16_t pad;       // inside a dim-4 group: 0 = a 4x4 source, 1 / 2 = a 2x2 source on the pair's first / second wire
  int32_t slot[3];
  int32_t const_off;
  uint32_t pad2;
};
static_assert(sizeof(BuildOp) == 24, "BuildOp layout");

// A matrix = product of build ops [begin,end) in tape order (later gate on the left).
struct BuildGroup {
  uint32_t begin, end;
  uint32_t mat_off;
  uint32_t dim;  // 2 or 4
};

// ---- fast tile path (k_tile2) ---------------------------------------------------------------
// A register-tile group whose LDS addressing is a host-built table: thread t gathers its 16
// amplitudes from byte addresses tbl[t] ^ off[c] (XOR-swizzle and the x8 already folded in).
// Basis permutations (X, CX: GF(2)-affine index maps) between groups never move data: they
// change the logical -> physical layout map the tables are built from.  `relayout` groups
// scatter to the identity layout (tbl2 / off2) behind an extra barrier: the last group of a
// stage does, so that the store / measure epilogue indexes the tile plainly.
// Ops of a group carry group-local bits (t0, c0 in 0..3) and a dispatch code in `pad`.
enum FastCode : uint8_t {
  FC_DENSE = 0,     // + tb                      dense 2x2 on bit tb
  FC_CDENSE = 4,    // + 3 * cb + (tb - (tb>cb)) controlled dense 2x2
  FC_DIAG = 16,     // + tb
  FC_CDIAG = 20,    // + 3 * cb + ..
  FC_X = 32,        // + tb                      in-register Pauli-X (pairs swapped)
  FC_CX = 36,       // + 3 * cb + ..             in-register CX
  FC_COUNT = 48
};
struct Group2 {
  uint32_t op_begin;   // first op in qmle_plan::ops2
  uint16_t n_ops;
  uint8_t relayout;
  uint8_t pad;
  uint32_t tbl;        // index into qmle_plan::tbl2 (one uint32 per thread of the workgroup)
  uint32_t tbl_out;    // relayout: scatter table
  uint32_t off[16];
  uint32_t off_out[16];
};
static_assert(sizeof(Group2) == 144, "Group2 layout");

enum StageKind : int { ST_DIRECT = 0, ST_TILE = 1, ST_DIAG_ALL = 2 };

struct Stage {
  int kind = ST_TILE;
  int op_begin = 0, op_end = 0;  // range in Plan::dev_ops
  int grp_begin = 0, grp_end = 0;  // range in Plan::op_groups (tile stages)
  int T = 0, L = 0;              // tile qubits, contiguous low bits
  int shift = 0;                 // > 0: the tile is the contiguous run of positions [shift, shift + T) (L = T; the
                                 // first stage of a top-first schedule: one tile per state, stored amplitude by amplitude)
  int n_tile_ops = 0;
  int8_t tile_bits[QMLE_MAX_QUBITS];   // ascending global positions of local bits
  int8_t outer_bits[QMLE_MAX_QUBITS];  // ascending global positions of the rest
  std::vector<int> src_ops;            // reference tape indices covered
  double algo_bytes_per_state = 0;     // SURVEY 8-d bytes of the covered gates
  // Known-zero tracking for runs that start from |0..0> (simulation.py:100): `zero_in` = bit
  // positions p such that every amplitude with bit p set is exactly zero when the stage
  // starts; `touched` = positions the stage's gates mix (non-diagonal targets), which leave
  // the set.  A tile stage never reads such amplitudes, and -- when the next stage is a tile
  // stage too (`next_tile`), which will not read them either -- never launches the tiles that
  // hold nothing else.
  uint32_t zero_in = 0, touched = 0;
  bool next_tile = false;
  // Every gate group sits on 4 bit positions of its own that are all known-zero on input: the
  // output tile is in[live bits] x prod_g (U_g e_0)[group bits] -- k_tile_product writes it
  // from the groups' first columns without staging amplitudes or running gates per tile.
  bool product_ok = false;
  // fast tile path: every gate is a (<= 1 control) 2x2 and T is in k_tile2's range
  bool fast_ok = false;
  int fast_begin = 0, fast_end = 0;  // range in qmle_plan::groups2
  uint32_t fast_gtab = 0;            // index into qmle_plan::tbl2: per-thread global byte offset of
                                     // the lane's first float4 inside the tile (k_tile2 prologue)
};

struct StageProfile {  // optional HIP-event timing of every stage launch (bench.py)
  bool on = false;
  std::vector<void *> start, stop;  // hipEvent_t pairs
  std::vector<int> stage;           // stage index per recorded pair
  size_t used = 0;
};

struct DevicePlan {  // lazily created by the first run on a device
  int device = -1;     // HIP device the image lives on; runs on another device are refused
  void *blob = nullptr;
  size_t blob_bytes = 0;
  LoweredOp *d_ops = nullptr;
  OpGroup *d_op_groups = nullptr;
  BuildOp *d_build = nullptr;
  BuildGroup *d_groups = nullptr;
  float *d_consts = nullptr;
  LoweredOp *d_ops2 = nullptr;   // fast tile path
  Group2 *d_groups2 = nullptr;
  uint32_t *d_tbl2 = nullptr;
};

}  // namespace qmle

struct qmle_plan {
  int n = 0, n_slots = 0;
  unsigned flags = 0;
  std::vector<qmle_op> ops;
  std::vector<float> consts;
  std::vector<qmle::LoweredOp> lowered;   // after 1-q merging, global positions
  std::vector<std::vector<int>> lowered_src;  // reference ops per lowered op
  std::vector<qmle::LoweredOp> dev_ops;   // per stage, stage-local positions
  std::vector<int> dev_src;               // source op of every dev_op (-1 if merged from several)
  std::vector<qmle::OpGroup> op_groups;   // register-tile groups of the tile stages
  std::vector<qmle::LoweredOp> ops2;      // fast tile path: ops of the Group2 groups
  std::vector<qmle::Group2> groups2;
  std::vector<uint32_t> tbl2;             // per-thread LDS byte addresses of the Group2 groups
  std::vector<qmle::BuildOp> build_ops;
  std::vector<qmle::BuildGroup> groups;   // needed-first: [0, n_groups_needed) are read by the forward tile / direct kernels
  int n_groups_needed = 0;
  std::vector<qmle::Stage> stages;
  uint32_t mat_floats = 0;                // per-sample matrix row length
  int fold_groups = 0;                    // most gate groups of any Stage::product_ok stage
  double model_cost = 0.0;                // pass-cost model of the chosen schedule (us per state at n = 24 scale)
  int chosen_candidate = -1;              // index of the schedule candidate the model picked (compile_plan)
  // plan autotuner (qmle_plan_autotune): a forced candidate / last-stage padding for this compile (-1: the
  // cost model resp. the QMLE_FORCE_CAND / QMLE_PAD_HIGH tuning switches), and every allowed candidate with
  // its model cost, cheapest first
  int force_candidate = -1, pad_high = -1;
  std::vector<std::pair<double, int>> cand_ranking;
  bool autotuned = false;
  // The same tape compiled for runs from |0..0> only (qmle_run_batch): its first stage may stage a
  // wider tile (it computes one tile per state whatever the size).  Owned; used by run_batch_masks
  // in place of this plan; qmle_apply_inplace / the adjoint sweep keep using this plan's stages.
  qmle_plan *zero_variant = nullptr;
  // complex128 engine (qmle_run_batch_f64): device copy of `lowered` + the constant blob as doubles
  // (lazily created); `consts64` optionally holds the caller's constants at full precision
  void *f64_blob = nullptr;
  int f64_device = -1;
  size_t n_user_consts = 0;               // constants handed to qmle_plan_create (the blob grows by permuted copies)
  std::vector<double> consts64;
  bool whole_state_lds = false;
  int tile_T = 0, tile_L = 0;
  double algo_bytes_per_state = 0;
  qmle::DevicePlan dev;
  qmle::StageProfile prof;
  // <Z> measurements only: trailing gates that map basis states to basis states (CX, SWAP)
  // or only add phases (diagonal gates) never touch a statevector -- they are folded into
  // the observables (Z_t -> Z_c Z_t under CX[c,t]) and `expval_child` runs the rest.
  qmle_plan *expval_child = nullptr;
  std::vector<qmle_op> absorbed;        // the folded gates, tape order
  double absorbed_algo_bytes = 0;       // their SURVEY 8-d bytes (credited to the last stage)
  double extra_algo_last_stage = 0;     // child side of the same number
  // adjoint sweep in LDS: device copies of the reverse tape (global positions) and its generator
  // terms, uploaded once per (plan, terms) -- owned by the REVERSE plan
  void *adj_blob = nullptr;
  uint64_t adj_hash = 0;
  // fused adjoint tile passes: per dev_op derivative index / generator type, per stage term
  // -> (gradient column, coefficient) tables
  void *adjf_blob = nullptr;
  uint64_t adjf_hash = 0;
};

// internal plan flag (bit 25; not part of the ABI): the plan is executed by qmle_run_batch only
#define QMLE_PLAN_INTERNAL_ZERO_RUN (1u << 25)

namespace qmle {
int compile_plan(qmle_plan *p);  // qmle_plan.cpp
// Split `ops` into the gates a <Z> measurement needs (`kept`) and the absorbable tail.
void split_expval_tail(const std::vector<qmle_op> &ops, int n, std::vector<qmle_op> &kept,
                       std::vector<qmle_op> &absorbed);
// Z on `wire` pulled back through the absorbed gates: bit w set <=> Z_w in the parity.
uint32_t pull_back_z(const std::vector<qmle_op> &absorbed, int wire);
std::string describe_plan(const qmle_plan *p);
// Which kernel measures <Z> / Z parities out of stage `si` when it is the last one of a run
// from |0..0>: 0 k_tile epilogue, 1 k_reg_measure<false>, 2 k_reg_measure<true> (gates folded
// into columns), 3 k_reg_measure_mono.  `sparse`: known-zero tracking is on for the run.
int expval_kernel_of(const qmle_plan *p, size_t si, bool sparse);
double algo_bytes(const qmle_op &op, int n);
constexpr int kFastMinT = 10, kFastMaxT = 13;  // k_tile2: 2^(T-4) threads, 8 float4 per thread
constexpr int kLdsMaxQubits = 14;       // 2^14 * 8 B = 128 KiB <= 160 KiB LDS/CU
constexpr int kDefaultTileBits = 13;    // 64 KiB tile -> 2 workgroups per CU
constexpr int kDefaultLowBits = 7;      // 128 amplitudes = 1 KiB contiguous per wave load
}  // namespace qmle
