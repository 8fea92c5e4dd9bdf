/*
 * qmle_sv.h -- C ABI of libqmle_sv.so, the MI355X (gfx950) statevector engine
 * that replaces the compute seam of cirKITers/qml-essentials' "jaqsi" simulator.
 *
 * The reference is 100 % Python/JAX and has no FFI of its own (SURVEY.md F2,
 * 8-b).  The seam this ABI replaces is
 *
 *   simulation.simulate_and_measure(tape, n_qubits, type, obs, use_density)
 *        qml_essentials/simulation.py:131-139     (per-sample kernel)
 *   Script._execute_batched(type, obs, args, kwargs, in_axes)
 *        qml_essentials/script.py:399-408          (jax.vmap over the batch;
 *        script.py:443-453 marks it as the multi-device replacement point)
 *
 * Conventions (identical to the reference):
 *   - state = 2^n complex64 (float2, re/im interleaved), wire 0 = MOST significant
 *     bit of the flat index            (simulation.py:100-104, test_jaqsi.py:416-427)
 *   - a k-qubit gate on wires [w0..w(k-1)] has matrix row/col index
 *     sum_j bit[w_j] << (k-1-j)        (operations.py:38-50)
 *   - every execution starts from |0...0>   (simulation.py:100)
 *   - batch element b uses row b of the angle table (jax.vmap semantics,
 *     script.py:302-315); results carry a leading batch dimension.
 *
 * Ownership: the caller (PyTorch-ROCm in this repo) owns every device buffer;
 * the library owns only the plan.  No exceptions cross the ABI: 0 = ok, negative
 * = qmle_status.  Plans are immutable after creation except for a lazily created
 * device copy of their descriptors; one stream per call; no other global state.
 * All pointers named d_* are device pointers, everything else is host memory.
 * Batch limits: qmle_run_batch / _parity / _f64 take any batch >= 1 (they cut it into launches
 * themselves); the stand-alone kernels on resident states (qmle_expval_z, qmle_probs,
 * qmle_marginal_probs, qmle_meyer_wallach, qmle_overlap, qmle_expval_parity, qmle_density*,
 * qmle_apply_inplace*, qmle_sample_counts, qmle_probs_diag_expval) take batch <= 65535 per call
 * (one grid row per sample), qmle_adjoint_gradient batch <= 32767 -- QMLE_ERR_INVALID_ARG above
 * that; the Python host side slices longer batches (qml-essentials_amd/_native.py).
 */
#ifndef QMLE_SV_H
#define QMLE_SV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QMLE_SV_VERSION 149 /* 0.1.4.8: qmle_adjoint_gradient_f64, k_direct_1q controlled-phase mode; 0.1.4.7: qmle_plan_executed (qmle_plan_expval_child = the folded child only); 0.1.4.6: qmle_plan_autotune; 0.1.4.5: QMLE_MEAS_MEYER_WALLACH; 0.1.4.4: qmle_philox_uniform_f32_device_key; 0.1.4.3: qmle_philox_uniform_f32_device; 0.1.4.2: qmle_apply_inplace_f64; 0.1.4.1: qmle_philox_uniform_f32 (host-side parameter sampler); 0.1.4: complex128 engine (qmle_run_batch_f64), qmle_meyer_wallach_reads, QMLE_ERR_INTERNAL; 0.1.3: fast tile path (no ABI change; a plan is bound to the device of its first run); 0.1.2: shot sampler; 0.1.1: qmle_op carries 4 wires (MAT4) */
#define QMLE_MAX_QUBITS 32

typedef struct qmle_plan qmle_plan;
typedef void *qmle_stream; /* hipStream_t */

typedef enum qmle_status {
  QMLE_OK = 0,
  QMLE_ERR_INVALID_ARG = -1,
  QMLE_ERR_WIRE_COUNT = -2,      /* operations.py:140-144  "expects k wire(s)" */
  QMLE_ERR_DUPLICATE_WIRES = -3, /* operations.py:145-146  "duplicate wires"   */
  QMLE_ERR_WIRE_RANGE = -4,      /* wire >= n_qubits                           */
  QMLE_ERR_UNKNOWN_OP = -5,
  QMLE_ERR_MEAS_TYPE = -6,       /* simulation.py:271 "Unknown measurement type" */
  QMLE_ERR_WORKSPACE = -7,       /* workspace too small                        */
  QMLE_ERR_HIP = -8,             /* a HIP runtime call failed                  */
  QMLE_ERR_NO_DEVICE = -9,
  QMLE_ERR_UNSUPPORTED = -10,
  QMLE_ERR_SLOT_RANGE = -11,
  QMLE_ERR_INTERNAL = -12        /* a build-time invariant of the kernels does not hold (e.g. a tile
                                    kernel whose dynamic LDS does not start at offset 0)          */
} qmle_status;

/* Gate vocabulary = the gate classes of qml_essentials/operations.py that the
 * 23 ansaetze, the encodings and the entanglement circuits put on the tape. */
typedef enum qmle_opcode {
  QMLE_OP_ID = 0,      /* operations.py:719  */
  QMLE_OP_X = 1,       /* PauliX  :746       */
  QMLE_OP_Y = 2,       /* PauliY  :762       */
  QMLE_OP_Z = 3,       /* PauliZ  :778       */
  QMLE_OP_H = 4,       /* :794               */
  QMLE_OP_S = 5,       /* :810               */
  QMLE_OP_RX = 6,      /* :1043  slot[0]=theta */
  QMLE_OP_RY = 7,      /* :1044              */
  QMLE_OP_RZ = 8,      /* :1045              */
  QMLE_OP_ROT = 9,     /* :1204  slot = phi,theta,omega */
  QMLE_OP_CX = 10,     /* :1098  wires = [control,target] */
  QMLE_OP_CY = 11,     /* :1099              */
  QMLE_OP_CZ = 12,     /* :1100              */
  QMLE_OP_CRX = 13,    /* :1485              */
  QMLE_OP_CRY = 14,    /* :1486              */
  QMLE_OP_CRZ = 15,    /* :1487              */
  QMLE_OP_CPHASE = 16, /* ControlledPhaseShift :1171 */
  QMLE_OP_SWAP = 17,   /* :831               */
  QMLE_OP_RXX = 18,    /* :1348              */
  QMLE_OP_RYY = 19,    /* :1349              */
  QMLE_OP_RZZ = 20,    /* :1350              */
  QMLE_OP_RZX = 21,    /* :1351              */
  QMLE_OP_CCX = 22,    /* :1103  wires = [c0,c1,target] */
  QMLE_OP_CSWAP = 23,  /* :1140  wires = [c,t0,t1]      */
  QMLE_OP_MAT1 = 24,   /* Operation(matrix=2x2), batch-constant; mat_off -> 8 floats  */
  QMLE_OP_MAT2 = 25,   /* Operation(matrix=4x4), batch-constant; mat_off -> 32 floats */
  QMLE_OP_DIAG_ALL = 26, /* DiagonalQubitUnitary on wires 0..n-1 in order (:922-926):
                            amp[i] *= exp(-i * consts[mat_off+i] * angle[slot0])
                            (Golomb encoding, unitary.py:661-701)              */
  QMLE_OP_MAT4 = 27,   /* dense 16x16 on 4 wires, batch-constant; mat_off -> 512 floats.
                          Superoperator of a 2-qubit Kraus channel on vec(rho)
                          (KrausChannel.apply_to_density, operations.py:1551-1578)   */
  QMLE_OP__COUNT = 28
} qmle_opcode;

/* One tape entry.  Barriers (operations.py:964) are not sent: simulate_pure skips
 * them (simulation.py:93-94). */
typedef struct qmle_op {
  uint16_t opcode;  /* qmle_opcode */
  int16_t wire[4];  /* reference wire order; unused = -1 */
  int32_t slot[3];  /* column of the angle table per parameter; unused = -1 */
  int32_t mat_off;  /* float offset into `consts` for MAT1/MAT2/DIAG_ALL, else -1 */
} qmle_op;

/* measure_state types (simulation.py:204-271) + engine-internal extras */
typedef enum qmle_meas {
  QMLE_MEAS_STATE = 0,    /* out: [B][2^n] complex64                           */
  QMLE_MEAS_PROBS = 1,    /* out: [B][2^n] float32  |psi|^2                    */
  QMLE_MEAS_EXPVAL_Z = 2, /* out: [B][n_obs] float32, PauliZ on obs_wires[k]
                             (fast path simulation.py:241-261, all in ONE pass) */
  QMLE_MEAS_DENSITY = 3,  /* out: [B][2^n][2^n] complex64 = |psi><psi|
                             (simulation.py:183-189)                           */
  QMLE_MEAS_MEYER_WALLACH = 4 /* out: [B][n + 1] float32 = (Q, Tr rho_w^2 of wire 0 .. n-1):
                             Meyer-Wallach of the state the plan produces
                             (entanglement.py:56-103 + jaqsi.py:60-103).  The plan's last tile
                             pass reports the sums of its own tile from LDS -- for n <= 14 that
                             is everything and no state is stored; above, the state is stored
                             and the remaining positions cost ceil((n - T) / 8) further reads
                             (2 at n = 28) instead of qmle_meyer_wallach's 3.  batch chunks of
                             <= 65535 states are handled inside. */
} qmle_meas;

/* plan flags */
#define QMLE_PLAN_DEFAULT 0u
#define QMLE_PLAN_NO_FUSION 1u      /* one HBM pass per reference gate (roofline mode) */
#define QMLE_PLAN_FORCE_GLOBAL 2u   /* never use the whole-state-in-LDS kernel        */
#define QMLE_PLAN_FORCE_TILE 4u     /* never use the direct per-gate kernels          */
#define QMLE_PLAN_NO_REGTILE 8u     /* one LDS sweep per gate inside a tile (debug/A-B) */
#define QMLE_PLAN_PREFETCH 16u      /* experiment: double-buffered LDS-DMA tile kernel (slower) */
#define QMLE_PLAN_NO_MERGE 64u      /* keep every 1-qubit gate its own operator (no RY.RZ.RY products):
                                       the fused adjoint sweep needs one generator per gate */
#define QMLE_PLAN_NO_ABSORB 32u     /* <Z>: simulate trailing CX / SWAP / diagonal gates instead of
                                       folding them into the observables (A-B, tests)   */
#define QMLE_PLAN_NO_SPARSE 128u    /* runs from |0..0>: read and store every amplitude, launch every tile,
                                       even where the state is known to be exactly zero (A-B, tests) */
#define QMLE_PLAN_TAPE_ORDER (1u << 24) /* schedule commuting gates in tape order instead of low bit
                                          positions first (A-B, tests)                       */
/* bits 8..15: tile qubits T override (0 = auto); bits 16..23: low-bit count L override */
#define QMLE_PLAN_TILE_BITS(t) (((unsigned)(t) & 0xffu) << 8)
#define QMLE_PLAN_LOW_BITS(l) (((unsigned)(l) & 0xffu) << 16)

int qmle_sv_version(void);
const char *qmle_status_string(int status);
/* number of HIP devices visible (0 if none / no driver); never fails */
int qmle_device_count(void);

/* Validate + compile a tape into an execution plan (host only, no HIP calls).
 * consts: batch-constant floats referenced by mat_off (copied). */
int qmle_plan_create(const qmle_op *ops, int n_ops, int n_qubits, int n_slots,
                     const float *consts, int n_consts, unsigned flags,
                     qmle_plan **out);
int qmle_plan_destroy(qmle_plan *plan);
/* Opt-in plan autotuner (the default schedule is the pass-cost model's choice, deterministically).
 * Times the model's best `top_k` schedule candidates (x both paddings of the last stage) for THIS
 * measurement and batch size on the current device -- `reps` runs each, zero angles, its own scratch
 * (hipMalloc'ed and freed inside: a one-off step, not part of a hot loop) -- and re-schedules the plan
 * that qmle_run_batch executes to the fastest (it must win by 1 %).  Candidates are the same tape under
 * another tile geometry / order of commuting gates: results agree to float32 rounding.  meas_type:
 * QMLE_MEAS_STATE or QMLE_MEAS_EXPVAL_Z.  chosen[2] (optional) <- candidate index, padding; ms_before /
 * ms_after (optional) <- per-call times of the old and the new schedule.  Plans with a single schedule
 * (whole state in LDS, forced geometry, QMLE_PLAN_NO_FUSION) return QMLE_OK untouched (chosen = -1, -1).
 * If the scratch allocation fails the plan keeps its schedule, QMLE_OK is returned and chosen[1] = -2
 * ("not tuned": the caller may try again later).  A choice is remembered per (executed plan's tape,
 * flags, batch class, device) for the life of the process: where both measurement types execute the
 * same plan the first one tuned decides for both.  Callers that cached qmle_workspace_bytes must ask
 * again afterwards. */
int qmle_plan_autotune(qmle_plan *plan, int meas_type, int n_obs, int batch, int top_k, int reps,
                       qmle_stream stream, int32_t *chosen, double *ms_before, double *ms_after);

/* The plan QMLE_MEAS_EXPVAL_Z actually executes: trailing gates that permute basis states
 * linearly (CX, SWAP) or are diagonal are folded into the Z observables (Z_t -> Z_c Z_t),
 * the remaining gates form this child plan (owned by `plan`; NULL if nothing was folded).
 * For introspection / profiling only. */
qmle_plan *qmle_plan_expval_child(qmle_plan *plan);
/* The plan object qmle_run_batch really executes for `meas_type` (owned by `plan`, never NULL for a
 * valid plan): the folded child above for QMLE_MEAS_EXPVAL_Z, and of that / of the plan itself the
 * schedule compiled for runs from |0..0> when there is one (wider first tile; such a handle is refused
 * by qmle_apply_inplace and qmle_adjoint_gradient, which apply stages to LIVE states:
 * QMLE_ERR_UNSUPPORTED).  Describe, count stages of and profile THIS handle. */
qmle_plan *qmle_plan_executed(qmle_plan *plan, int meas_type);
/* JSON description of the compiled passes (for tests / DESIGN.md); returns the
 * number of bytes needed (excluding NUL); writes at most cap-1 bytes + NUL. */
int qmle_plan_describe(const qmle_plan *plan, char *buf, size_t cap);
/* counts: [0]=reference gates, [1]=HBM passes, [2]=whole-state-LDS(0/1),
 * [3]=tile qubits T, [4]=floats of per-sample matrices, [5]=direct passes */
int qmle_plan_stats(const qmle_plan *plan, int64_t stats[8]);

/* Minimum workspace for `batch` samples and the bytes that let the engine keep
 * `states_in_flight` states resident (0 = engine default).  A batch larger than one chunk of that many
 * states is run with TWO chunks in flight, one stage apart, on two streams the library owns (they fork from
 * `stream` by an event and join it again before the call returns its last launch: to the caller the call is
 * ordered on `stream` like any other); the figure returned includes the second set of state buffers.  A
 * smaller workspace is legal: the engine splits what it gets in two, or keeps one chunk on `stream` alone.
 * QMLE_NO_CHUNK_OVERLAP=1 (read per call): always the one-stream loop. */
size_t qmle_workspace_bytes(const qmle_plan *plan, int batch, int meas_type,
                            int n_obs, int states_in_flight);

/* simulate_and_measure over a batch: d_angles is [batch][n_slots] float32,
 * d_out as documented per qmle_meas.  obs_wires is HOST memory.  */
int qmle_run_batch(qmle_plan *plan, const float *d_angles, int batch, int meas_type,
                   const int32_t *obs_wires, int n_obs, void *d_out, void *d_workspace,
                   size_t workspace_bytes, qmle_stream stream);
/* Same, for Z (x) Z (x) .. parity observables (jaqsi.py:149-167; Model with tuple
 * output_qubit, model.py:980-990): wire_masks[k] (HOST) has bit w set iff wire w takes part.
 * Measured like QMLE_MEAS_EXPVAL_Z -- out of the last pass, no stored state; d_out
 * float32 [batch][n_obs]; workspace as for QMLE_MEAS_EXPVAL_Z. */
int qmle_run_batch_parity(qmle_plan *plan, const float *d_angles, int batch,
                          const uint32_t *wire_masks, int n_obs, float *d_out, void *d_workspace,
                          size_t workspace_bytes, qmle_stream stream);

/* Angle table from DEVICE-resident arguments: out[b][s] = const[s] + sum_t coef[t] *
 * leaf_{arg[t]}[row_k(b)][idx[t]], terms of slot s = [ptr[s], ptr[s+1]); row_k(b) =
 * ((b + batch_offset) / div_k) % mod_k  (cartesian inputs x params batch, model.py:1449-1481).
 * d_leaves / strides / div / mod are HOST arrays of n_leaves <= 8 entries; everything d_* is
 * device memory.  The sum runs in fp64; d_period (may be NULL) holds one double per slot: where
 * > 0 the angle is reduced into (-period/2, period/2] before the float32 store (4 pi for the
 * rotation gates, which depend on angle / 2 only -- binary / ternary encodings scale inputs
 * by up to 3^(n-1), ansaetze/encodings of model.py:804-816).  Replaces the host-side angle
 * arithmetic when params / inputs already live in HBM. */
int qmle_build_angles(const float *const *d_leaves, const int64_t *leaf_strides,
                      const int32_t *leaf_div, const int32_t *leaf_mod, int n_leaves,
                      const int32_t *d_ptr, const int32_t *d_arg, const int32_t *d_idx,
                      const float *d_coef, const float *d_const, const double *d_period,
                      int n_slots, int64_t batch, int64_t batch_offset, float *d_out,
                      qmle_stream stream);

/* qmle_build_angles + qmle_run_batch behind ONE call (the Model's device route: its angle map is all the
 * host holds, model.py:804-816 / ansaetze.py:323-371).  `map` carries qmle_build_angles' arguments;
 * d_angles [batch][plan n_slots] float32 is SCRATCH: the call may fill and read it (fewer than 64 samples, or
 * a plan with the Golomb diagonal, which reads its angle in the pass itself) or leave it untouched -- from 64
 * samples on the per-sample gate matrices are built straight from the map (same arithmetic, same results
 * bit for bit, one kernel and the table's round trip less).  May be NULL for a plan without slots.  The
 * host's time between the two launches -- 8-14 us of an idle GPU in the 0.2 ms analysis loops of BASELINE
 * configs 3 / 4 -- is gone either way. */
typedef struct qmle_angle_map {
  const float *const *d_leaves;   /* HOST array of n_leaves device pointers */
  const int64_t *leaf_strides;    /* HOST */
  const int32_t *leaf_div;        /* HOST */
  const int32_t *leaf_mod;        /* HOST */
  int32_t n_leaves;
  const int32_t *d_ptr, *d_arg, *d_idx;
  const float *d_coef, *d_const;
  const double *d_period;         /* may be NULL */
  int64_t batch_offset;
} qmle_angle_map;
int qmle_run_batch_map(qmle_plan *plan, const qmle_angle_map *map, float *d_angles, int batch,
                       int meas_type, const int32_t *obs_wires, int n_obs, void *d_out,
                       void *d_workspace, size_t workspace_bytes, qmle_stream stream);

/* Apply the plan's passes in place to resident states [batch][2^n] complex64 (no
 * initialisation, no measurement) -- the per-gate loop simulation.py:102-103 alone.
 * Workspace: qmle_workspace_bytes(plan, batch, QMLE_MEAS_STATE, 0, 0). */
int qmle_apply_inplace(qmle_plan *plan, const float *d_angles, int batch, void *d_states,
                       void *d_workspace, size_t workspace_bytes, qmle_stream stream);

/* ---- complex128 execution (the reference's jax_enable_x64 mode, operations.py:12-16; switched on
 * by tests/test_coefficients.py:19, test_entanglement.py:13, test_ansaetze.py:18) ----------------
 * Same plan, same tape: angles float64 [batch][n_slots]; out = [B][2^n] complex128 (STATE),
 * [B][2^n] float64 (PROBS), [B][n_obs] float64 (EXPVAL_Z; wire_masks[k] has bit w set iff wire w
 * takes part in the Z (x) Z ... observable k -- HOST array), [B][2^n][2^n] complex128 (DENSITY,
 * n <= 12).  Every gate of the tape is applied (no observable folding, no known-zero shortcuts):
 * the accuracy mode, 1e-10 against a complex128 reference.  n <= 13: the state stays in LDS for
 * the whole circuit; above, one streaming pass per (merged) operator. */
int qmle_run_batch_f64(qmle_plan *plan, const double *d_angles, int batch, int meas_type,
                       const uint32_t *wire_masks, int n_obs, void *d_out, void *d_workspace,
                       size_t workspace_bytes, qmle_stream stream);
size_t qmle_workspace_bytes_f64(const qmle_plan *plan, int batch, int meas_type);
/* optional: the batch-constant blob of qmle_plan_create at full precision (explicit matrices of a
 * complex128 caller); same length, before the first qmle_run_batch_f64 of the plan */
int qmle_plan_set_consts_f64(qmle_plan *plan, const double *consts, int n_consts);
/* complex128 counterpart of qmle_apply_inplace: the plan's operators on resident states
 * [batch][2^n] complex128 (batch <= 65535 per call), no initialisation, no measurement.  The
 * doubled-register density-matrix path uses it between Kraus channels in x64 mode
 * (simulation.py:107-128). */
int qmle_apply_inplace_f64(qmle_plan *plan, const double *d_angles, int batch, void *d_states,
                           void *d_workspace, size_t workspace_bytes, qmle_stream stream);
size_t qmle_apply_inplace_f64_workspace_bytes(const qmle_plan *plan, int batch);

/* Optional per-pass HIP-event timing (bench.py's live roofline measurement): between
 * begin and end every pass launch of this plan is bracketed by an event pair on the
 * launch stream.  end() synchronises, fills stage_ms[i] / stage_launches[i] (i = pass
 * index of qmle_plan_describe) and returns 1 if `capacity` pairs were not enough. */
int qmle_profile_begin(qmle_plan *plan, int capacity);
int qmle_profile_end(qmle_plan *plan, double *stage_ms, int64_t *stage_launches, int n_stages);

/* ---- stand-alone measurement / analysis kernels on resident states ---------- */
/* d_states: [batch][2^n] complex64 */
int qmle_expval_z(const void *d_states, int n_qubits, int batch, const int32_t *obs_wires,
                  int n_obs, float *d_out, void *d_workspace, size_t workspace_bytes,
                  qmle_stream stream);
size_t qmle_expval_workspace_bytes(int n_qubits, int batch);
int qmle_probs(const void *d_states, int n_qubits, int batch, float *d_out,
               qmle_stream stream);
int qmle_density(const void *d_states, int n_qubits, int batch, void *d_out,
                 qmle_stream stream);
/* probs marginalised onto `keep` wires, kept wires in ascending wire order
 * (jaqsi.py:106-146): d_out [batch][2^n_keep] float32 (zeroed by the call) */
int qmle_marginal_probs(const void *d_states, int n_qubits, int batch,
                        const int32_t *keep_wires, int n_keep, float *d_out,
                        qmle_stream stream);
/* F_i = |<psi_i|psi_{i+n_pairs}>|^2, i < n_pairs; d_states holds 2*n_pairs states
 * (Expressibility pairing rule, expressibility.py:49-52; pure-state form of
 * math.py:60-86) */
int qmle_pair_fidelity(const void *d_states, int n_qubits, int n_pairs, float *d_out,
                       void *d_workspace, size_t workspace_bytes, qmle_stream stream);
size_t qmle_pair_fidelity_workspace_bytes(int n_qubits, int n_pairs);
/* Measurements of a vectorised density matrix (2n-"qubit" register, ket wires first:
 * the layout of Operation.apply_to_density, operations.py:485-512): diagonal
 * probabilities [batch][2^n] (measure_density "probs", simulation.py:303-304) and <Z> on
 * obs_wires from that diagonal. */
int qmle_density_probs(const void *d_rho, int n_qubits, int batch, float *d_out,
                       qmle_stream stream);
int qmle_density_expval_z(const void *d_rho, int n_qubits, int batch, const int32_t *obs_wires,
                          int n_obs, float *d_out, qmle_stream stream);
/* <a_i|b_i> for i < count (complex64 out[count]); the matrix-free general
 * observable path: <O> = Re <psi | O psi>  (simulation.py:263-269) */
int qmle_overlap(const void *d_a, const void *d_b, int n_qubits, int count, void *d_out,
                 void *d_workspace, size_t workspace_bytes, qmle_stream stream);
size_t qmle_overlap_workspace_bytes(int n_qubits, int count);
/* Z (x) Z (x) ... parity observables (jaqsi.py:149-167): wire_masks[k] has bit w set
 * iff wire w takes part in observable k (HOST array); d_out [batch][n_obs] float32 */
int qmle_expval_parity(const void *d_states, int n_qubits, int batch, const uint32_t *wire_masks,
                       int n_obs, float *d_out, void *d_workspace, size_t workspace_bytes,
                       qmle_stream stream);
size_t qmle_expval_parity_workspace_bytes(int n_qubits, int batch);
/* Meyer-Wallach: d_out[b] = 2 (1 - 1/n sum_j Tr rho_j^2), Tr rho_j^2 = a^2+d^2+2|c|^2
 * (entanglement.py:86-101 via the Schmidt identity).
 * d_purities (optional, may be NULL): [batch][n] float32 */
int qmle_meyer_wallach(const void *d_states, int n_qubits, int batch, float *d_out,
                       float *d_purities, void *d_workspace, size_t workspace_bytes,
                       qmle_stream stream);
size_t qmle_meyer_wallach_workspace_bytes(int n_qubits, int batch);
/* how many times qmle_meyer_wallach streams the state from HBM at this size (host only;
 * bench.py's bytes-moved accounting): the passes of the tile scheme from 12 qubits on, one
 * cache-resident sweep per wire below */
int qmle_meyer_wallach_reads(int n_qubits);
/* Host only, no device work: out[i] = the i-th value of
 * numpy.random.Generator(numpy.random.Philox(key=key)).uniform(low, high, n).astype(float32),
 * bit for bit (Philox4x64-10, numpy's counter and double conventions).  The parameter sampler
 * behind Model.initialize_params -- the reference's jax.random.uniform (model.py:687-693) runs
 * on a threefry stream that cannot be reproduced here; the build's stream is
 * numpy's Philox, and numpy's own loop (~9 ns per value) was two thirds of the wall-clock of
 * Expressibility(12 q, 1024 pairs). */
int qmle_philox_uniform_f32(const uint64_t key[2], uint64_t n, double low, double high, float *out);
/* The same stream written by the GPU into d_out (one work item per Philox block; bit for bit the
 * host version's floats): what utils.uniform uses when a GPU is present. */
int qmle_philox_uniform_f32_device(const uint64_t key[2], uint64_t n, double low, double high,
                                   float *d_out, qmle_stream stream);
/* The same with the key read from DEVICE memory (d_key: two 64-bit words) when the kernel runs: a
 * launch captured in a hipGraph is re-seeded by rewriting those words, not by re-capturing. */
int qmle_philox_uniform_f32_device_key(const uint64_t *d_key, uint64_t n, double low, double high,
                                       float *d_out, qmle_stream stream);
/* numpy.histogram(values, bins=linspace(lo,hi,n_bins+1)) counts (last bin
 * right-inclusive) -- expressibility.py:104-108; d_counts int32[n_bins], zeroed
 * by the call */
int qmle_histogram(const float *d_values, int64_t count, int n_bins, float lo, float hi,
                   int32_t *d_counts, qmle_stream stream);

/* Shot sampling (simulation.py:320-377 sample_shots): for every row of d_probs
 * [batch][2^n] float32 draw `shots` basis states by inverse-CDF sampling (the algorithm of
 * jax.random.choice(key, dim, (shots,), p=probs): r = total * u, first index with
 * cumsum >= r) and histogram them.  u comes from Philox4x32-10 keyed by `seed`, counter =
 * (shot pair, row_offset + row) -- results do not depend on how a batch is chunked or
 * sharded.  d_counts int32 [batch][2^n] (zeroed by the call); d_est_probs (optional, may be
 * NULL) float32 counts / shots.  Workspace: the fp64 CDF. */
int qmle_sample_counts(const float *d_probs, int n_qubits, int batch, int shots, uint64_t seed,
                       uint64_t row_offset, int32_t *d_counts, float *d_est_probs,
                       void *d_workspace, size_t workspace_bytes, qmle_stream stream);
size_t qmle_sample_workspace_bytes(int n_qubits, int batch);
/* sum_i p[b][i] * diag(O_k lifted)[i] for n_obs observables (simulation.py:363-372, the
 * computational-basis estimate Tr(O diag(p))).  HOST arrays: obs_wires = concatenated wire
 * lists, obs_n_wires[k] wires each (first wire = most significant bit of the table index),
 * obs_diag_off[k] = float offset of the 2^k-entry diagonal in the DEVICE array d_diag, or
 * -1 for a Z (x) Z (x) .. parity.  d_out float32 [batch][n_obs]. */
int qmle_probs_diag_expval(const float *d_probs, int n_qubits, int batch,
                           const int32_t *obs_wires, const int32_t *obs_n_wires,
                           const int32_t *obs_diag_off, const float *d_diag, int n_obs,
                           float *d_out, void *d_workspace, size_t workspace_bytes,
                           qmle_stream stream);
size_t qmle_probs_diag_expval_workspace_bytes(int n_obs);

/* ---- adjoint differentiation ---------------------------------------------------------------
 * Gradient of  C = sum_b sum_k weights[b][k] <Z..Z>_k(b)  with respect to every gate angle in
 * ONE backward sweep (what jax.grad through Script.execute gives the reference,
 * tests/test_jaqsi.py:131-141, tests/test_model.py:1097-1145, docs/training.md):
 *   psi = U_N .. U_1 |0>,  lambda = (sum_k w_k Z_k) psi;  for k = N .. 1:
 *   dC/dtheta_k = coef_k Im <lambda| G_k |psi>,  then  psi <- U_k^+ psi,  lambda <- U_k^+ lambda.
 * `fwd` is the ordinary plan of the tape; `rev` is a QMLE_PLAN_NO_FUSION | QMLE_PLAN_NO_ABSORB
 * plan of the REVERSED, DAGGERED tape (one gate per pass), `terms[r]` describes the
 * generator of rev op r:  G = i^n_y X^{x_wires} Z^{z_wires} projected onto |1> on
 * proj_wires (controls), or diag(consts[marks_off + i]) for the Golomb diagonal.
 * d_grad float32 [batch][n_grad_slots] (zeroed by the call); out_slot < 0 = no derivative. */
typedef struct qmle_adjoint_term {
  int32_t out_slot;
  uint32_t x_wires, z_wires, proj_wires; /* bit w = wire w */
  int32_t n_y;
  float coef;
  int32_t marks_off;
} qmle_adjoint_term;
int qmle_adjoint_gradient(qmle_plan *fwd, qmle_plan *rev, const float *d_angles_fwd,
                          const float *d_angles_rev, int batch, const float *d_weights,
                          const uint32_t *obs_wire_masks, int n_obs,
                          const qmle_adjoint_term *terms, int n_terms, float *d_grad,
                          int n_grad_slots, void *d_workspace, size_t workspace_bytes,
                          qmle_stream stream);
size_t qmle_adjoint_workspace_bytes(const qmle_plan *fwd, const qmle_plan *rev, int batch);
/* The same sweep on the complex128 engine (x64 mode: jax.grad with jax_enable_x64,
 * /root/reference/tests/test_jaqsi.py:57,131-141,764-786): angles, weights and gradients are float64,
 * psi and lambda complex128, one streaming launch per operator (no fusion, like qmle_run_batch_f64).
 * `rev` keeps one source gate per operator (QMLE_PLAN_NO_FUSION or QMLE_PLAN_NO_MERGE); any batch size
 * (samples are processed in rounds of <= 4 GiB of psi + lambda).  d_grad float64 [batch][n_grad_slots]. */
int qmle_adjoint_gradient_f64(qmle_plan *fwd, qmle_plan *rev, const double *d_angles_fwd,
                              const double *d_angles_rev, int batch, const double *d_weights,
                              const uint32_t *obs_wire_masks, int n_obs,
                              const qmle_adjoint_term *terms, int n_terms, double *d_grad,
                              int n_grad_slots, void *d_workspace, size_t workspace_bytes,
                              qmle_stream stream);
size_t qmle_adjoint_workspace_bytes_f64(const qmle_plan *fwd, const qmle_plan *rev, int batch);

#ifdef __cplusplus
}
#endif
#endif /* QMLE_SV_H */
