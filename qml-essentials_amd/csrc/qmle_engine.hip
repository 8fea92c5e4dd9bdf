// libqmle_sv, engine: plan objects on the device, the angle table / gate matrix builders, and the
// C-ABI entry points that run a plan (qmle_run_batch, qmle_apply_inplace, profiling).
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
#include "qmle_matrices.h"

#include <map>
#include <mutex>
#include <vector>

namespace {

template <bool GMAJOR>
__global__ void __launch_bounds__(64)
k_build_matrices(const BuildOp *__restrict__ build, const BuildGroup *__restrict__ groups, int n_groups,
                 const float *__restrict__ angles, int n_slots, const float *__restrict__ consts,
                 float *__restrict__ mats, uint32_t mat_floats, int batch) {
  build_matrices_body<const float *, float, float, GMAJOR>(build, groups, n_groups, angles, n_slots, consts, mats, mat_floats, batch);
}

// the same straight from the affine angle map (qmle_run_batch_map; whole waves per group only)
__global__ void __launch_bounds__(64)
k_build_matrices_map(const BuildOp *__restrict__ build, const BuildGroup *__restrict__ groups, int n_groups,
                     const AngleMapSrc map, int n_slots, const float *__restrict__ consts,
                     float *__restrict__ mats, uint32_t mat_floats, int batch) {
  build_matrices_body<const AngleMapSrc &, float, float, true>(build, groups, n_groups, map, n_slots, consts, mats,
                                                              mat_floats, batch);
}

// ---------------------------------------------------------------------------
// angle table from device-resident leaves:  table[b][s] = c[s] + sum_t coef[t] * leaf_{arg[t]}[row][idx[t]]
// (gate angles are affine in params / inputs, ansaetze.py:323-371, model.py:804-816); the row
// of leaf k for flattened sample b is (b / div_k) % mod_k -- the cartesian batch of
// model.py:1449-1481 without materialising the repeats.
// ---------------------------------------------------------------------------
// One work item per (state, slot), flattened (every lane busy whatever the slot count: a block-per-state
// layout with the divisions hoisted out of the term loop measured 6.5 instead of 4.1 us for the 2048 x 108
// table of the Expressibility loop -- the kernel is a chain of three dependent loads, not arithmetic).
// IDX: 32-bit index arithmetic when batch x slots and batch + offset allow, 64-bit otherwise.
template <class IDX>
__global__ void __launch_bounds__(256)
k_build_angles(AngleLeaves lv, const int *__restrict__ ptr, const int *__restrict__ arg,
               const int *__restrict__ idx, const float *__restrict__ coef,
               const float *__restrict__ cst, const double *__restrict__ period, int n_slots,
               long long batch, long long b_offset, float *__restrict__ out) {
  const IDX i = (IDX)blockIdx.x * 256u + threadIdx.x;
  if ((long long)i >= batch * n_slots) return;
  const IDX b = i / (IDX)n_slots;
  const int s = (int)(i - b * (IDX)n_slots);
  const IDX gb = b + (IDX)b_offset;
  // fp64 accumulation, then reduction into (-period/2, period/2] (4 pi for a rotation gate, which
  // depends on angle / 2 only): binary / ternary encodings scale inputs by up to 3^(n-1), and a
  // float32 angle of ~1500 rad would carry 1e-4 rad of rounding into the gate matrices
  double acc = (double)cst[s];
  for (int t = ptr[s]; t < ptr[s + 1]; ++t) {
    const int k = arg[t];
    const long long row = (long long)((gb / (IDX)lv.div[k]) % (IDX)lv.mod[k]);
    acc = fma((double)coef[t], (double)lv.ptr[k][row * lv.stride[k] + idx[t]], acc);
  }
  const double per = period ? period[s] : 0.0;
  if (per > 0.0 && (acc > per || acc < -per)) acc -= per * rint(acc / per);
  out[i] = (float)acc;
}

}  // namespace

namespace qmle {

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

// A run that starts from |0..0> keeps track of the amplitudes that are still exactly zero
// (Stage::zero_in); the prefetching experiment does not.
bool plan_sparse(const qmle_plan *p) {
  static const bool pf_env = std::getenv("QMLE_PREFETCH") != nullptr;
  return !(p->flags & (QMLE_PLAN_NO_SPARSE | QMLE_PLAN_PREFETCH)) && !pf_env;
}

// qmle_run_batch_map hands its angle map to the matrix builder of the batch it is about to run (this
// thread's next launch_build_matrices of >= 64 samples): the angles are then formed inside the builder and
// the table is never written
static thread_local const AngleMapSrc *tls_angle_map = nullptr;

// per-sample gate matrices for the whole batch: d_mats[b][mat_floats] from d_angles[b][n_slots]
int launch_build_matrices(const qmle_plan *p, const float *d_angles, float *d_mats, int batch, hipStream_t stream,
                          bool forward_only) {
  const int ng = forward_only ? p->n_groups_needed : (int)p->groups.size();
  if (ng <= 0) return QMLE_OK;
  if (tls_angle_map && batch >= 64) {
    const uint64_t waves = (uint64_t)ng * (((uint64_t)batch + 63) / 64);
    hipLaunchKernelGGL(k_build_matrices_map, dim3(grid_for(waves * 64, 64)), dim3(64), 0, stream, p->dev.d_build,
                       p->dev.d_groups, ng, *tls_angle_map, p->n_slots, p->dev.d_consts, d_mats, p->mat_floats, batch);
    HIPCHK(hipGetLastError());
    return QMLE_OK;
  }
  // one work item per (sample, group); from 64 samples on, whole waves per group (build_matrices_body)
  const uint64_t items = batch >= 64 ? (uint64_t)ng * (((uint64_t)batch + 63) / 64) * 64 : (uint64_t)batch * (uint64_t)ng;
  if (batch >= 64)
    hipLaunchKernelGGL(k_build_matrices<true>, dim3(grid_for(items, 64)), dim3(64), 0, stream, p->dev.d_build,
                       p->dev.d_groups, ng, d_angles, p->n_slots, p->dev.d_consts, d_mats, p->mat_floats, batch);
  else
    hipLaunchKernelGGL(k_build_matrices<false>, dim3(grid_for(items, 64)), dim3(64), 0, stream, p->dev.d_build,
                       p->dev.d_groups, ng, d_angles, p->n_slots, p->dev.d_consts, d_mats, p->mat_floats, batch);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

}  // namespace qmle

// ===========================================================================
// C ABI
// ===========================================================================
// (every qmle_* function below is declared extern "C" by include/qmle_sv.h; the definitions inherit it)

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
// the child that runs when trailing gates were folded into the Z observables (NULL: nothing was folded)
qmle_plan *qmle_plan_expval_child(qmle_plan *plan) { return plan ? plan->expval_child : nullptr; }
// the plan qmle_run_batch really executes for `meas_type`: the folded child for QMLE_MEAS_EXPVAL_Z when
// there is one, and of that (or of the plan) the schedule compiled for runs from |0..0> when the
// pass-cost model preferred it.  Never NULL for a valid plan; describe / profile THIS handle.
qmle_plan *qmle_plan_executed(qmle_plan *plan, int meas_type) {
  if (!plan) return nullptr;
  qmle_plan *p = (meas_type == QMLE_MEAS_EXPVAL_Z && plan->expval_child) ? plan->expval_child : plan;
  return p->zero_variant ? p->zero_variant : p;
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
size_t qmle::ws_mats_bytes(const qmle_plan *p, int batch) {
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

static size_t expval_partial_rows(const qmle_plan *p) {
  size_t rows = (size_t)expval_blocks(p->n);
  if ((size_t)overlap_blocks(p->n) > rows) rows = (size_t)overlap_blocks(p->n);
  if (!p->stages.empty() && p->stages.back().kind == ST_TILE && !p->whole_state_lds) {
    const size_t tiles = (size_t)1 << (p->n - p->stages.back().T);
    if (tiles > rows) rows = tiles;
  }
  return rows;
}

// Meyer-Wallach measurement: can the plan's last pass report its tile's sums (tile_mw_row)?
// Policy (measured on MI355X): when the last pass holds the WHOLE state (n <= 14) the sums cost
// one epilogue and no statevector is ever stored -- always taken.  For tiled states the round-4 epilogue (208
// packed fmas + a 41-value wave reduction per work item and tile) cost the pass what the saved read was worth
// (n = 28: 0.99 ms after the circuit either way) and stayed opt-in; since round 5 the producing pass leaves positions
// 0..3 to the first later read (lean epilogue) and streams its stores, so that the later reads do not run into its
// write-back: 0.79 ms after the circuit against 0.92-1.0 for the stand-alone reads -- taken by default
// (QMLE_MW_FUSE_TILED=0: the stand-alone reads; read per call: A/B, tests).
static bool plan_mw_fusable(const qmle_plan *p, bool whatever_the_switches = false) {
  if (p->stages.empty()) return false;
  const Stage &last = p->stages.back();
  if (!whatever_the_switches) {
    if (std::getenv("QMLE_NO_MW_FUSION") != nullptr) return false;
    // (round 5: tiled states fuse by default -- lean epilogue + streaming stores, n = 28: 0.81 ms after the circuit
    // against 1.0 ms for the three stand-alone reads; QMLE_MW_FUSE_TILED=0 keeps the stand-alone reads: A/B, tests)
    const char *ft = std::getenv("QMLE_MW_FUSE_TILED");
    if (last.T < p->n && ft && atoi(ft) == 0) return false;
  }
  return mw_fusable(p->n, last);
}
// rows + purities of `batch` states (conservative per state: the rows per state shrink with the batch)
static size_t mw_ws_bytes(const qmle_plan *p, int batch) {
  // (enough for either route: the switches are read per call, a workspace sized before one was flipped stays valid)
  size_t one = mw_resident_ws_bytes(p->n, 1);
  if (plan_mw_fusable(p, true)) one = std::max(one, mw_fused_ws_bytes(p->n, 1, p->stages.back()));
  return align_up(one, 256) * (size_t)batch;
}

// Round 5: consecutive chunks of a batch are independent (own states, own partial sums), and their passes want
// different things from the GPU -- the zero fill HBM writes, the one-tile-per-state kernel 32 workgroups, a measuring
// pass the vector unit.  Two processes sharing the card ran the all-live K2 step 18 % faster than one (1.93 against 1.63 M
// gate-applies/s); the same overlap inside one call: chunks alternate between two internal streams that fork from the
// caller's stream and join it again.  QMLE_NO_CHUNK_OVERLAP=1: the one-stream loop (read per call).
static bool chunk_overlap_on() { return std::getenv("QMLE_NO_CHUNK_OVERLAP") == nullptr; }
constexpr int kPipeStages = 32;
struct SideStreams {
  hipStream_t s[2] = {nullptr, nullptr};
  hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
  // software pipeline: stage k of chunk i + 1 starts when stage k of chunk i is done -- two chunks in lockstep
  // (fill next to fill, measuring pass next to measuring pass) share the card evenly and gain nothing; one stage
  // apart, the HBM-bound pass of one chunk runs beside the vector-bound pass of the other
  hipEvent_t stage_done[2][kPipeStages] = {};
  bool ok = false;
};
static SideStreams *side_streams() {
  static SideStreams per_dev[kMaxDevices];
  static std::mutex mu;
  const int dev = current_device();
  if (dev < 0 || dev >= kMaxDevices) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  SideStreams &x = per_dev[dev];
  if (!x.ok) {
    bool good = hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < 2 && good; ++k) {
      good = hipStreamCreateWithFlags(&x.s[k], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&x.join[k], hipEventDisableTiming) == hipSuccess;
      for (int e = 0; e < kPipeStages && good; ++e)
        good = hipEventCreateWithFlags(&x.stage_done[k][e], hipEventDisableTiming) == hipSuccess;
    }
    if (!good) { (void)hipGetLastError(); return nullptr; }
    x.ok = true;
  }
  return &x;
}

static size_t per_state_ws_bytes(const qmle_plan *p, int meas_type) {
  size_t b = align_up((size_t)8 << p->n, 256);
  if (meas_type == QMLE_MEAS_EXPVAL_Z)
    b += align_up(expval_partial_rows(p) * (QMLE_MAX_QUBITS + 1) * sizeof(float), 256);
  if (meas_type == QMLE_MEAS_MEYER_WALLACH) b += mw_ws_bytes(p, 1);
  return b;
}

size_t qmle::workspace_bytes_one(const qmle_plan *plan, int batch, int meas_type,
                                 int states_in_flight) {
  size_t total = ws_mats_bytes(plan, batch) + 512;  // + alignment slack
  const bool lds_direct_meas =
      plan->whole_state_lds && (meas_type == QMLE_MEAS_PROBS || meas_type == QMLE_MEAS_EXPVAL_Z);
  if (meas_type == QMLE_MEAS_MEYER_WALLACH && plan->whole_state_lds && plan_mw_fusable(plan))
    return total + mw_ws_bytes(plan, batch);  // the sums come out of the LDS tile: no state is stored
  if (meas_type != QMLE_MEAS_STATE && !lds_direct_meas) {
    int s = states_in_flight > 0 ? states_in_flight : default_states_in_flight(plan, batch);
    if (s > batch) s = batch;
    // (a batch of several chunks runs them alternately on two internal streams, each with its own state / partial
    // buffers -- chunk_overlap below; a caller that hands in less gets the one-stream loop)
    const int slots = (s < batch && chunk_overlap_on()) ? 2 : 1;
    total += (size_t)slots * (size_t)s * per_state_ws_bytes(plan, meas_type);
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
  if (meas_type < QMLE_MEAS_STATE || meas_type > QMLE_MEAS_MEYER_WALLACH) return QMLE_ERR_MEAS_TYPE;
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
int qmle::run_batch_masks(qmle_plan *plan, const float *d_angles, int batch, int meas_type,
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
  rc = launch_build_matrices(plan, d_angles, d_mats, batch, stream, /*forward_only=*/true);
  if (rc != QMLE_OK) return rc;

  const size_t D = (size_t)1 << n;
  const size_t sb = D * sizeof(float2);

  // ---- whole state in LDS: one launch does simulate + measure ------------------
  if (plan->whole_state_lds && meas_type == QMLE_MEAS_MEYER_WALLACH && plan_mw_fusable(plan)) {
    // circuit + Meyer-Wallach sums in one launch per chunk: the state never leaves the LDS
    const Stage &st = plan->stages[0];
    for (int b0 = 0; b0 < batch; b0 += 65535) {
      const int bc = batch - b0 < 65535 ? batch - b0 : 65535;
      if (workspace_bytes < mw_ws_bytes(plan, bc)) return QMLE_ERR_WORKSPACE;
      ProfScope prof_scope(plan, 0, stream);
      rc = launch_tile(plan, st, nullptr, d_mats + (size_t)b0 * plan->mat_floats,
                       d_angles ? d_angles + (size_t)b0 * plan->n_slots : nullptr, bc, true, TM_MW_ONLY, ws,
                       nullptr, 0, stream);
      if (rc == QMLE_OK)
        rc = run_mw_fused(nullptr, n, bc, st, 0, ws, workspace_bytes, (float *)d_out + (size_t)b0 * (n + 1), stream);
      if (rc != QMLE_OK) return rc;
    }
    return QMLE_OK;
  }
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
  // two slots (state buffer + partial sums each) when the batch has several chunks and the workspace has room
  const float2 *const d_states0_dbg = d_states;
  SideStreams *side = nullptr;
  char *slot1 = nullptr;
  // (one streaming pass per gate -- QMLE_PLAN_NO_FUSION -- is HBM-bound in every pass: two of them at once share the
  // bandwidth and lose 2-3 % to the mix; those plans keep the one-stream loop)
  if (in_flight < batch && chunk_overlap_on() && !(plan->flags & QMLE_PLAN_NO_FUSION)) {
    if (meas_type == QMLE_MEAS_STATE) {
      // every chunk writes its own rows of d_out; partial sums are not used
      side = side_streams();
    } else {
      // (`workspace_bytes` is what was left when the state buffers were carved: they start at d_states.  A workspace
      // sized by qmle_workspace_bytes holds two slots of the chunk it was asked for; one that holds less is split in two)
      const size_t per = per_state_ws_bytes(plan, meas_type);
      int two = (int)std::min<size_t>(workspace_bytes / per / 2, 65535);
      const int dflt = default_states_in_flight(plan, batch);
      if (two > dflt) two = dflt;
      if (two >= 1 && (side = side_streams()) != nullptr) {
        // the partial sums sit behind the state block of `in_flight` states: re-carve slot 0 for `two`
        in_flight = two;
        slot1 = (char *)d_states + (size_t)two * per;
        ws = (char *)d_states + (size_t)two * align_up(sb, 256);
      }
    }
  }
  void *d_partial = ws;
  const size_t partial_bytes =
      (size_t)in_flight * expval_partial_rows(plan) * (QMLE_MAX_QUBITS + 1) * sizeof(float);
  // <Z> straight out of the last tile pass (no store of the final state, no extra read)
  const bool fuse_expval = meas_type == QMLE_MEAS_EXPVAL_Z && !plan->stages.empty() &&
                           plan->stages.back().kind == ST_TILE;
  const bool fuse_mw = meas_type == QMLE_MEAS_MEYER_WALLACH && plan_mw_fusable(plan);
  const size_t mw_bytes = meas_type == QMLE_MEAS_MEYER_WALLACH ? mw_ws_bytes(plan, in_flight) : 0;

  if (std::getenv("QMLE_DBG_OVERLAP"))
    fprintf(stderr, "[qmle] chunks of %d of %d states, overlap %s (slot1 %p, ws %zu bytes, used before the states %zu, two slots %zu)\n",
            in_flight, batch, side ? "on" : "off", (void *)slot1, workspace_bytes,
            (size_t)((char *)d_states0_dbg - (char *)d_workspace), 2 * (size_t)in_flight * per_state_ws_bytes(plan, meas_type));
  hipStream_t const caller_stream = stream;
  if (side) {  // fork: what the caller's stream has queued so far (the matrices, the angle table) comes first
    if (hipEventRecord(side->fork, caller_stream) != hipSuccess ||
        hipStreamWaitEvent(side->s[0], side->fork, 0) != hipSuccess ||
        hipStreamWaitEvent(side->s[1], side->fork, 0) != hipSuccess) {
      (void)hipGetLastError();
      side = nullptr;
    }
  }
  float2 *const d_states0 = d_states;
  void *const d_partial0 = d_partial;
  auto join_side = [&]() {
    if (!side) return;
    for (int k = 0; k < 2; ++k)
      if (hipEventRecord(side->join[k], side->s[k]) == hipSuccess)
        (void)hipStreamWaitEvent(caller_stream, side->join[k], 0);
  };
  struct JoinGuard {  // (also on the error returns inside the loop)
    decltype(join_side) &f;
    ~JoinGuard() { f(); }
  } join_guard{join_side};
  int chunk_no = 0;
  for (int b0 = 0; b0 < batch; b0 += in_flight, ++chunk_no) {
    const int bc = batch - b0 < in_flight ? batch - b0 : in_flight;
    if (side) {
      stream = side->s[chunk_no & 1];
      if (slot1 && (chunk_no & 1)) {
        d_states = (float2 *)slot1;
        d_partial = slot1 + ((char *)d_partial0 - (char *)d_states0);
      } else {
        d_states = d_states0;
        d_partial = d_partial0;
      }
    }
    float2 *stc = meas_type == QMLE_MEAS_STATE ? d_states + (size_t)b0 * D : d_states;
    const float *mats = d_mats + (size_t)b0 * plan->mat_floats;
    const float *ang = d_angles ? d_angles + (size_t)b0 * plan->n_slots : nullptr;
    bool initialised = false;
    int reg_q = -1;  // >= 0: the last pass ran as k_reg_measure with 2^reg_q tiles per row
    int tile_row_shift = 0;  // k_tile2's multi-tile measuring variant: 2^shift tiles per row
    for (size_t si = 0; si < plan->stages.size(); ++si) {
      const Stage &st = plan->stages[si];
      const bool piped = side && plan->stages.size() <= (size_t)kPipeStages;
      if (piped && chunk_no > 0)  // behind the same stage of the previous chunk (the other stream)
        (void)hipStreamWaitEvent(stream, side->stage_done[(chunk_no - 1) & 1][si], 0);
      struct StageDone {  // recorded when the stage's launches are queued (any exit from this iteration)
        hipEvent_t ev; hipStream_t st_;
        ~StageDone() { if (ev) (void)hipEventRecord(ev, st_); }
      } stage_done{piped ? side->stage_done[chunk_no & 1][si] : nullptr, stream};
      ProfScope prof_scope(plan, (int)si, stream);
      if (st.kind == ST_TILE && fuse_mw && si + 1 == plan->stages.size()) {
        // the last pass stores the state AND reports its tile's Meyer-Wallach sums (one row per tile)
        rc = launch_tile(plan, st, stc, mats, ang, bc, !initialised, TM_STORE_MW, d_partial, nullptr, 0, stream,
                         /*from_zero=*/true, d_cols ? d_cols + (size_t)b0 * plan->fold_groups * 16 : nullptr,
                         &tile_row_shift);
        initialised = true;
      } else if (st.kind == ST_TILE) {
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
                         last_fused && (single_bits || tm == TM_EXPVAL_MASKS) ? &tile_row_shift : nullptr);
        initialised = true;
      } else {
        if (!initialised) {
          launch_init_zero(stc, n, bc, stream);
          initialised = true;
        }
        if (st.kind == ST_DIRECT) {
          rc = launch_direct(plan, plan->dev_ops[st.op_begin], stc, mats, bc, stream);
        } else {
          const LoweredOp &o = plan->dev_ops[st.op_begin];
          launch_diag_all(stc, n, bc, plan->dev.d_consts + o.mat_off, ang, plan->n_slots, o.slot, stream);
          rc = QMLE_OK;
        }
      }
      if (rc != QMLE_OK) return rc;
    }
    if (!initialised) launch_init_zero(stc, n, bc, stream);
    // measure this chunk
    if (meas_type == QMLE_MEAS_PROBS) {
      const uint64_t tc = (uint64_t)bc * (D / 2);
      launch_probs(stc, (float *)d_out + (size_t)b0 * D, tc, stream);
    } else if (meas_type == QMLE_MEAS_EXPVAL_Z && fuse_expval) {
      ObsBits ob;  // column of the 33-float row: the bit's sum, or (masks) the observable's own
      const bool by_position = (single_bits || semi_single) && reg_q < 0;
      for (int k = 0; k < QMLE_MAX_QUBITS; ++k) ob.row_mask[k] = 0u;
      for (int k = 0; k < n_obs; ++k) {
        ob.bits[k] = by_position ? obs_bits[k] : (int8_t)k;
        if (by_position && semi_single) ob.row_mask[k] = row_masks[k];
      }
      const int tiles = (1 << (n - plan->stages.back().T)) >> (reg_q < 0 ? tile_row_shift : reg_q);
      launch_expval_final((const float *)d_partial, tiles, bc, n_obs, ob, (float *)d_out + (size_t)b0 * n_obs, stream);
    } else if (meas_type == QMLE_MEAS_EXPVAL_Z) {
      rc = single_bits
               ? run_expval(stc, n, bc, obs_bits, n_obs, (float *)d_out + (size_t)b0 * n_obs,
                            d_partial, partial_bytes, stream)
               : run_parity_pos(stc, n, bc, obs_masks, n_obs, (float *)d_out + (size_t)b0 * n_obs,
                                d_partial, partial_bytes, stream);
      if (rc != QMLE_OK) return rc;
    } else if (meas_type == QMLE_MEAS_MEYER_WALLACH) {
      rc = fuse_mw ? run_mw_fused(stc, n, bc, plan->stages.back(), tile_row_shift, d_partial, mw_bytes,
                                  (float *)d_out + (size_t)b0 * (n + 1), stream)
                   : run_mw_resident(stc, n, bc, d_partial, mw_bytes, (float *)d_out + (size_t)b0 * (n + 1), stream);
      if (rc != QMLE_OK) return rc;
    } else if (meas_type == QMLE_MEAS_DENSITY) {
      if (n > 15) return QMLE_ERR_UNSUPPORTED;
      launch_density(stc, (float2 *)d_out + (size_t)b0 * D * D, n, bc, stream);
    }
    HIPCHK(hipGetLastError());
  }
  return QMLE_OK;
}

// Apply the plan's passes IN PLACE to resident states (no |0..0> initialisation, no
// measurement): the gate-application hot loop on its own (simulation.py:102-103).
// One stage of a plan applied in place to resident states.
int qmle::run_stage_inplace(qmle_plan *plan, const Stage &st, float2 *d_states, const float *d_mats,
                            const float *d_angles, int batch, hipStream_t stream) {
  if (st.kind == ST_TILE)
    return launch_tile(plan, st, d_states, d_mats, d_angles, batch, false, TM_STORE, nullptr,
                       nullptr, 0, stream);
  if (st.kind == ST_DIRECT)
    return launch_direct(plan, plan->dev_ops[st.op_begin], d_states, d_mats, batch, stream);
  const LoweredOp &o = plan->dev_ops[st.op_begin];
  launch_diag_all(d_states, plan->n, batch, plan->dev.d_consts + o.mat_off, d_angles, plan->n_slots, o.slot, stream);
  return QMLE_OK;
}

int qmle_apply_inplace(qmle_plan *plan, const float *d_angles, int batch, void *d_states,
                       void *d_workspace, size_t workspace_bytes, qmle_stream stream_) {
  if (!plan || batch < 1 || batch > 65535 || !d_states || !d_workspace) return QMLE_ERR_INVALID_ARG;
  if (plan->n_slots > 0 && !d_angles) return QMLE_ERR_INVALID_ARG;
  // (a schedule compiled for runs from |0..0> -- qmle_plan_executed's handle -- is not one for live states)
  if (plan->flags & QMLE_PLAN_INTERNAL_ZERO_RUN) return QMLE_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  int rc = ensure_device_plan(plan);
  if (rc != QMLE_OK) return rc;
  char *ws = (char *)d_workspace;
  const size_t mis = (size_t)(256 - ((uintptr_t)ws & 255)) & 255;
  if (workspace_bytes < mis + ws_mats_bytes(plan, batch)) return QMLE_ERR_WORKSPACE;
  float *d_mats = (float *)(ws + mis);
  rc = launch_build_matrices(plan, d_angles, d_mats, batch, stream, /*forward_only=*/true);
  if (rc != QMLE_OK) return rc;
  int stage_idx = -1;
  for (const Stage &st : plan->stages) {
    ProfScope prof_scope(plan, ++stage_idx, stream);
    rc = run_stage_inplace(plan, st, (float2 *)d_states, d_mats, d_angles, batch, stream);
    if (rc != QMLE_OK) return rc;
  }
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}


// ---- plan autotuner (opt-in; the default schedule stays the cost model's, deterministically) ----------
// The pass-cost model ranks the schedule candidates from measurements at n = 24; which positions share a
// tile moves a pass by up to +-20 % through an address hash nobody documents, and away from
// n = 24 the model's favourite is sometimes 5-7 % off the best.  qmle_plan_autotune times the model's best
// `top_k` candidates (x the two paddings of the last stage) ON THE DEVICE with the caller's batch size and
// measurement, and re-schedules the plan that qmle_run_batch executes to the fastest.  Every candidate is
// the same tape under another order of commuting gates / another tile geometry: results equal to float32
// rounding (tests/test_gpu_kernels.py runs all 48).  Choices are remembered per (tape, flags, device).
namespace {

struct TunedChoice { int cand, pad; };
static std::mutex g_tuned_mu;
static std::map<uint64_t, TunedChoice> g_tuned;

static uint64_t tape_hash(const qmle_plan *p, int meas_class, int batch_class) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void *data, size_t bytes) {
    const unsigned char *c = (const unsigned char *)data;
    for (size_t i = 0; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; }
  };
  mix(p->ops.data(), p->ops.size() * sizeof(qmle_op));
  const int v[5] = {p->n, p->n_slots, (int)p->flags, meas_class, batch_class};
  mix(v, sizeof(v));
  const int dev = current_device();
  mix(&dev, sizeof(dev));
  return h;
}

// a stand-alone compile of `base`'s tape with a forced candidate (no variant, no child)
static qmle_plan *compile_candidate(const qmle_plan *base, int cand, int pad) {
  qmle_plan *c = new (std::nothrow) qmle_plan();
  if (!c) return nullptr;
  c->n = base->n;
  c->n_slots = base->n_slots;
  c->flags = base->flags;
  c->ops = base->ops;
  c->consts.assign(base->consts.begin(), base->consts.begin() + base->n_user_consts);
  c->n_user_consts = base->n_user_consts;
  c->force_candidate = cand;
  c->pad_high = pad;
  c->extra_algo_last_stage = base->extra_algo_last_stage;
  if (compile_plan(c) != QMLE_OK || c->chosen_candidate != cand) {
    (void)qmle_plan_destroy(c);
    return nullptr;
  }
  return c;
}

static bool same_stages(const qmle_plan *a, const qmle_plan *b) {
  if (a->stages.size() != b->stages.size()) return false;
  for (size_t s = 0; s < a->stages.size(); ++s)
    if (a->stages[s].T != b->stages[s].T || a->stages[s].kind != b->stages[s].kind ||
        std::memcmp(a->stages[s].tile_bits, b->stages[s].tile_bits, (size_t)a->stages[s].T) != 0 ||
        a->stages[s].op_end - a->stages[s].op_begin != b->stages[s].op_end - b->stages[s].op_begin)
      return false;
  return true;
}

// dst keeps its identity (handles, children, folded tail); its schedule becomes src's
static void adopt_schedule(qmle_plan *dst, qmle_plan *src) {
  if (dst->dev.blob) { (void)hipFree(dst->dev.blob); dst->dev = DevicePlan(); }
  if (dst->f64_blob) { (void)hipFree(dst->f64_blob); dst->f64_blob = nullptr; dst->f64_device = -1; }
  dst->consts.swap(src->consts);
  dst->lowered.swap(src->lowered);
  dst->lowered_src.swap(src->lowered_src);
  dst->dev_ops.swap(src->dev_ops);
  dst->dev_src.swap(src->dev_src);
  dst->op_groups.swap(src->op_groups);
  dst->ops2.swap(src->ops2);
  dst->groups2.swap(src->groups2);
  dst->tbl2.swap(src->tbl2);
  dst->build_ops.swap(src->build_ops);
  dst->groups.swap(src->groups);
  dst->stages.swap(src->stages);
  dst->cand_ranking.swap(src->cand_ranking);
  std::swap(dst->mat_floats, src->mat_floats);
  std::swap(dst->n_groups_needed, src->n_groups_needed);
  std::swap(dst->fold_groups, src->fold_groups);
  std::swap(dst->model_cost, src->model_cost);
  std::swap(dst->chosen_candidate, src->chosen_candidate);
  std::swap(dst->whole_state_lds, src->whole_state_lds);
  std::swap(dst->tile_T, src->tile_T);
  std::swap(dst->tile_L, src->tile_L);
  std::swap(dst->algo_bytes_per_state, src->algo_bytes_per_state);
  dst->force_candidate = src->force_candidate;
  dst->pad_high = src->pad_high;
  dst->autotuned = true;
}

}  // namespace

int qmle_plan_autotune(qmle_plan *plan, int meas_type, int n_obs, int batch, int top_k, int reps,
                       qmle_stream stream_, int32_t *chosen, double *ms_before, double *ms_after) {
  if (!plan || batch < 1 || top_k < 1 || reps < 1) return QMLE_ERR_INVALID_ARG;
  if (meas_type != QMLE_MEAS_STATE && meas_type != QMLE_MEAS_EXPVAL_Z) return QMLE_ERR_MEAS_TYPE;
  if (meas_type == QMLE_MEAS_EXPVAL_Z && (n_obs < 1 || n_obs > plan->n)) return QMLE_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (chosen) { chosen[0] = chosen[1] = -1; }
  if (ms_before) *ms_before = 0.0;
  if (ms_after) *ms_after = 0.0;
  // the plan this measurement executes
  qmle_plan *owner = (meas_type == QMLE_MEAS_EXPVAL_Z && plan->expval_child) ? plan->expval_child : plan;
  qmle_plan *target = owner->zero_variant ? owner->zero_variant : owner;
  if (target->whole_state_lds || target->cand_ranking.size() < 2 || (target->flags & (QMLE_PLAN_NO_FUSION | QMLE_PLAN_PREFETCH)) ||
      ((target->flags >> 8) & 0xffffu))
    return QMLE_OK;  // one schedule only: nothing to tune
  const int batch_class = batch >= 256 ? 2 : batch >= 16 ? 1 : 0;
  // keyed by the TARGET plan, not by the measurement: without a folded child "state" and "expval" execute the
  // same plan, and two remembered choices for one plan would overwrite each other's schedule (and rebuild the
  // device image) on every alternation.  The first measurement tuned decides; the other adopts its choice.
  const uint64_t key = tape_hash(target, owner != plan ? 1 : 0, batch_class);
  {
    std::lock_guard<std::mutex> lock(g_tuned_mu);
    auto it = g_tuned.find(key);
    if (it != g_tuned.end()) {  // tuned before in this process: adopt the remembered choice, no timing
      if (it->second.cand != target->chosen_candidate || it->second.pad != (target->pad_high > 0 ? 1 : 0)) {
        qmle_plan *c = compile_candidate(target, it->second.cand, it->second.pad);
        if (c) { adopt_schedule(target, c); (void)qmle_plan_destroy(c); }
      }
      target->autotuned = true;
      if (chosen) { chosen[0] = target->chosen_candidate; chosen[1] = target->pad_high > 0 ? 1 : 0; }
      return QMLE_OK;
    }
  }
  // scratch of this one-off step: a zero angle table, the output and a workspace large enough for every candidate
  const int n = target->n;
  const size_t D = (size_t)1 << n;
  const size_t ang_bytes = (size_t)batch * (target->n_slots ? target->n_slots : 1) * sizeof(float);
  const size_t out_bytes = meas_type == QMLE_MEAS_STATE ? (size_t)batch * D * sizeof(float2) : (size_t)batch * n_obs * sizeof(float);
  uint32_t masks[QMLE_MAX_QUBITS];
  for (int k = 0; k < QMLE_MAX_QUBITS; ++k) masks[k] = 1u << (k < n ? k : 0);
  struct Cand { qmle_plan *p; int cand, pad; double ms; };
  std::vector<Cand> cands;
  cands.push_back({nullptr, target->chosen_candidate, target->pad_high > 0 ? 1 : 0, 0.0});  // the current schedule, as it stands
  for (size_t i = 0; i < target->cand_ranking.size() && (int)i < top_k; ++i)
    for (int pad = 0; pad < 2; ++pad) {
      const int k = target->cand_ranking[i].second;
      if (k == cands[0].cand && pad == cands[0].pad) continue;
      qmle_plan *c = compile_candidate(target, k, pad);
      if (!c) continue;
      // (a padding that changes nothing compiles to the same stages: skip the duplicate)
      bool dup = false;
      for (const Cand &o : cands)
        if (o.cand == k && same_stages(o.p ? o.p : target, c)) dup = true;
      if (dup) { (void)qmle_plan_destroy(c); continue; }
      cands.push_back({c, k, pad, 0.0});
    }
  size_t ws_bytes = workspace_bytes_one(target, batch, meas_type, 0);
  for (const Cand &c : cands)
    if (c.p) ws_bytes = std::max(ws_bytes, workspace_bytes_one(c.p, batch, meas_type, 0));
  char *scratch = nullptr;
  const size_t total = align_up(ang_bytes, 256) + align_up(out_bytes, 256) + ws_bytes + 512;
  if (hipMalloc((void **)&scratch, total) != hipSuccess) {
    for (Cand &c : cands) if (c.p) (void)qmle_plan_destroy(c.p);
    (void)hipGetLastError();
    if (chosen) chosen[1] = -2;  // "not tuned" (as opposed to -1 / -1: nothing to tune): the caller may try again
    return QMLE_OK;  // no room to tune next to the caller's buffers: keep the model's schedule
  }
  float *d_ang = (float *)scratch;
  void *d_out = scratch + align_up(ang_bytes, 256);
  void *d_ws = scratch + align_up(ang_bytes, 256) + align_up(out_bytes, 256);
  int rc = QMLE_OK;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipMemsetAsync(d_ang, 0, ang_bytes, stream) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
      hipEventCreate(&e1) != hipSuccess)
    rc = QMLE_ERR_HIP;
  for (size_t i = 0; i < cands.size() && rc == QMLE_OK; ++i) {
    qmle_plan *run = cands[i].p ? cands[i].p : target;
    for (int r = -1; r < reps && rc == QMLE_OK; ++r) {  // r = -1: warm-up (device image upload, attributes)
      if (r == 0) (void)hipEventRecord(e0, stream);
      rc = run_batch_masks(run, d_ang, batch, meas_type, masks, n_obs, d_out, d_ws, ws_bytes, stream);
    }
    if (rc != QMLE_OK) break;
    (void)hipEventRecord(e1, stream);
    if (hipEventSynchronize(e1) != hipSuccess) { rc = QMLE_ERR_HIP; break; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    cands[i].ms = (double)ms / reps;
  }
  size_t best = 0;
  if (rc == QMLE_OK) {
    for (size_t i = 1; i < cands.size(); ++i)
      if (cands[i].ms < cands[best].ms * 0.99) best = i;  // (a candidate must win by 1 %: ties keep the model's choice)
    if (ms_before) *ms_before = cands[0].ms;
    if (ms_after) *ms_after = cands[best].ms;
    if (best != 0) adopt_schedule(target, cands[best].p);
    target->autotuned = true;
    if (chosen) { chosen[0] = cands[best].cand; chosen[1] = cands[best].pad; }
    std::lock_guard<std::mutex> lock(g_tuned_mu);
    g_tuned[key] = {cands[best].cand, cands[best].pad};
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipStreamSynchronize(stream);
  (void)hipFree(scratch);
  for (Cand &c : cands) if (c.p) (void)qmle_plan_destroy(c.p);
  return rc;
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
    // (a leaf without elements -- the parameters of an ansatz that has none -- may come as a NULL pointer: no term can refer to it)
    if ((!d_leaves[k] && leaf_strides[k] != 0) || leaf_div[k] < 1 || leaf_mod[k] < 1) return QMLE_ERR_INVALID_ARG;
    lv.ptr[k] = d_leaves[k];
    lv.stride[k] = leaf_strides[k];
    lv.div[k] = leaf_div[k];
    lv.mod[k] = leaf_mod[k];
  }
  const uint64_t total = (uint64_t)batch * (uint64_t)n_slots;
  if (total + 256 < (1ull << 32) && batch_offset >= 0 && (uint64_t)batch + (uint64_t)batch_offset < (1ull << 32))
    hipLaunchKernelGGL(k_build_angles<uint32_t>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, lv,
                       d_ptr, d_arg, d_idx, d_coef, d_const, d_period, n_slots, (long long)batch,
                       (long long)batch_offset, d_out);
  else
    hipLaunchKernelGGL(k_build_angles<unsigned long long>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       lv, d_ptr, d_arg, d_idx, d_coef, d_const, d_period, n_slots, (long long)batch,
                       (long long)batch_offset, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_run_batch_map(qmle_plan *plan, const qmle_angle_map *map, float *d_angles, int batch,
                       int meas_type, const int32_t *obs_wires, int n_obs, void *d_out,
                       void *d_workspace, size_t workspace_bytes, qmle_stream stream) {
  if (!plan || !map || batch < 1) return QMLE_ERR_INVALID_ARG;
  if (plan->n_slots == 0)
    return qmle_run_batch(plan, d_angles, batch, meas_type, obs_wires, n_obs, d_out, d_workspace, workspace_bytes,
                          stream);
  if (!d_angles) return QMLE_ERR_INVALID_ARG;
  // The table is needed as a table only by the Golomb diagonal (QMLE_OP_DIAG_ALL reads its angle in the pass
  // itself); every other gate meets its angles in the matrix builder, which can form them from the map --
  // one kernel (5-9 us of the 0.17 ms analysis loops) and the table's round trip less.  From 64 samples on
  // (whole waves per group); QMLE_NO_MAP_FUSION=1 keeps the two kernels (A/B, tests).
  bool table = batch < 64 || std::getenv("QMLE_NO_MAP_FUSION") != nullptr || map->n_leaves < 0 || map->n_leaves > 8 ||
               !map->d_ptr || !map->d_const;
  for (const qmle_op &o : plan->ops) table = table || o.opcode == QMLE_OP_DIAG_ALL;
  if (table) {
    const int rc = qmle_build_angles(map->d_leaves, map->leaf_strides, map->leaf_div, map->leaf_mod, map->n_leaves,
                                     map->d_ptr, map->d_arg, map->d_idx, map->d_coef, map->d_const, map->d_period,
                                     plan->n_slots, batch, map->batch_offset, d_angles, stream);
    if (rc != QMLE_OK) return rc;
    return qmle_run_batch(plan, d_angles, batch, meas_type, obs_wires, n_obs, d_out, d_workspace, workspace_bytes,
                          stream);
  }
  AngleMapSrc src;
  std::memset(&src, 0, sizeof(src));
  for (int k = 0; k < map->n_leaves; ++k) {
    if ((!map->d_leaves[k] && map->leaf_strides[k] != 0) || map->leaf_div[k] < 1 || map->leaf_mod[k] < 1) return QMLE_ERR_INVALID_ARG;
    src.lv.ptr[k] = map->d_leaves[k];
    src.lv.stride[k] = map->leaf_strides[k];
    src.lv.div[k] = map->leaf_div[k];
    src.lv.mod[k] = map->leaf_mod[k];
  }
  src.ptr = map->d_ptr; src.arg = map->d_arg; src.idx = map->d_idx;
  src.coef = map->d_coef; src.cst = map->d_const; src.period = map->d_period;
  src.b_offset = map->batch_offset;
  src.small = map->batch_offset >= 0 && (uint64_t)batch + (uint64_t)map->batch_offset < (1ull << 32);
  tls_angle_map = &src;
  const int rc = qmle_run_batch(plan, d_angles, batch, meas_type, obs_wires, n_obs, d_out, d_workspace,
                                workspace_bytes, stream);
  tls_angle_map = nullptr;
  return rc;
}

int qmle_profile_begin(qmle_plan *plan, int capacity) {
  if (!plan || capacity < 1) return QMLE_ERR_INVALID_ARG;
  // (qmle_run_batch executes the from-|0..0> variant when there is one: its launches are the ones to time)
  if (plan->zero_variant) {
    const int rc = qmle_profile_begin(plan->zero_variant, capacity);
    if (rc != QMLE_OK) return rc;
  }
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
  if (!plan || !stage_ms || !stage_launches) return QMLE_ERR_INVALID_ARG;
  if (plan->zero_variant && plan->zero_variant->prof.on) {
    // the variant ran (qmle_run_batch) unless the plan itself recorded launches (qmle_apply_inplace)
    if (plan->prof.used == 0) {
      plan->prof.on = false;
      for (size_t k = 0; k < plan->prof.start.size(); ++k) {
        (void)hipEventDestroy((hipEvent_t)plan->prof.start[k]);
        (void)hipEventDestroy((hipEvent_t)plan->prof.stop[k]);
      }
      plan->prof.start.clear(); plan->prof.stop.clear(); plan->prof.stage.clear();
      return qmle_profile_end(plan->zero_variant, stage_ms, stage_launches, n_stages);
    }
    std::vector<double> ms(plan->zero_variant->stages.size() + 1);
    std::vector<int64_t> cnt(plan->zero_variant->stages.size() + 1);
    (void)qmle_profile_end(plan->zero_variant, ms.data(), cnt.data(), (int)ms.size());
  }
  if (n_stages < (int)plan->stages.size()) return QMLE_ERR_INVALID_ARG;
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

