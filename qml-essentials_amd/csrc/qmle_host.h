// Host-side interfaces between the translation units of libqmle_sv (not part of the ABI):
//   qmle_engine.hip    plan objects on the device, angle / matrix builders, qmle_run_batch & co.
//   qmle_tile.hip      LDS-tile passes (k_tile, k_tile2, k_reg_measure*, product passes)
//   qmle_direct.hip    streaming passes (one gate in place, the Golomb diagonal, fills)
//   qmle_analysis.hip  measurement / analysis kernels of resident states, samplers
//   qmle_adjoint.hip   adjoint differentiation
//   qmle_f64.hip       complex128 engine
// Kernels stay private to their unit (anonymous namespaces); what crosses a unit boundary is a
// plain host function that launches them.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "qmle_internal.h"

namespace qmle {

// Which column of a 33-float partial row an observable reads, and the sign of row i:
// (-1)^popcount(row_mask[k] & i) -- observables that are Z on ONE position of the last tile times
// Z's on outer positions (bits of the tile index).
struct ObsBits {
  int8_t bits[QMLE_MAX_QUBITS] = {};
  uint32_t row_mask[QMLE_MAX_QUBITS] = {};
};

// what a tile pass does with the finished tile (launch_tile's `meas`)
enum TileMeas : int {
  TM_STORE = 0,   // write the tile back into the state buffer
  TM_PROBS = 1,   // write |psi|^2 to out (float)
  TM_EXPVAL = 2,  // whole-state only: <Z> on obs bits
  TM_EXPVAL_PARTIAL = 3,  // last pass of a tiled state: per-tile signed sums for EVERY bit
                          // -> out[b][tile][33] (k_expval_final reduces); state not stored
  TM_EXPVAL_MASKS = 4,    // same, for Z-parity observables (obs_mask): per-tile Walsh-Hadamard
                          // transform of |psi|^2 -> out[b][tile][k < n_obs]
  TM_STORE_MW = 5,        // TM_STORE + the tile's Meyer-Wallach sums -> out[b][tile][kMwFusedRow] (tile_mw_row)
  TM_MW_ONLY = 6,         // whole-state tiles: the Meyer-Wallach row alone, no state is stored
};

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

// ---- qmle_engine.hip ----
int ensure_device_plan(qmle_plan *p);
bool plan_sparse(const qmle_plan *p);  // known-zero tracking is on for runs of this plan
size_t ws_mats_bytes(const qmle_plan *p, int batch);
size_t workspace_bytes_one(const qmle_plan *plan, int batch, int meas_type, int states_in_flight);
// forward_only: just the matrices the forward tile / direct kernels read (qmle_plan::n_groups_needed)
int launch_build_matrices(const qmle_plan *p, const float *d_angles, float *d_mats, int batch, hipStream_t stream,
                          bool forward_only = false);
int run_batch_masks(qmle_plan *plan, const float *d_angles, int batch, int meas_type,
                    const uint32_t *obs_masks, int n_obs, void *d_out, void *d_workspace,
                    size_t workspace_bytes, hipStream_t stream);
int run_stage_inplace(qmle_plan *plan, const Stage &st, float2 *d_states, const float *d_mats,
                      const float *d_angles, int batch, hipStream_t stream);

// ---- qmle_tile.hip ----
size_t tile_lds_bytes(int T, int L, int n_slots);
int tile_threads(int T);
int launch_tile(const qmle_plan *p, const Stage &st, float2 *states, const float *mats,
                const float *angles, int batch, bool init_zero, int meas, void *out,
                const uint32_t *obs_masks, int n_obs, hipStream_t stream, bool from_zero = false,
                float2 *cols = nullptr, int *row_shift = nullptr);
int reg_measure_kind(const qmle_plan *p, size_t si, int n_obs);
int launch_reg_measure(const qmle_plan *p, const Stage &st, int kind, float2 *states,
                       const float *mats, const float *angles, int batch, void *out,
                       const uint32_t *obs_masks, int n_obs, hipStream_t stream, int *q_out,
                       float *coef);

// ---- qmle_direct.hip ----
int launch_direct(const qmle_plan *p, const LoweredOp &op, float2 *states, const float *mats,
                  int batch, hipStream_t stream);
void launch_init_zero(float2 *states, int n, int batch, hipStream_t stream);   // |0..0> per state
void launch_diag_all(float2 *states, int n, int batch, const float *marks, const float *angles,
                     int n_slots, int slot, hipStream_t stream);
void launch_fill_zero(float2 *states, uint64_t n_float4, hipStream_t stream);

// ---- qmle_analysis.hip ----
int expval_blocks(int n);
int overlap_blocks(int n);
int run_expval(const float2 *states, int n, int batch, const int8_t *obs_bits, int n_obs,
               float *d_out, void *ws, size_t ws_bytes, hipStream_t stream);
int run_parity_pos(const float2 *states, int n, int batch, const uint32_t *pos_masks, int n_obs,
                   float *d_out, void *ws, size_t ws_bytes, hipStream_t stream);
void launch_expval_final(const float *partial, int n_rows, int batch, int n_obs, const ObsBits &ob,
                         float *d_out, hipStream_t stream);
void launch_probs(const float2 *states, float *d_out, uint64_t total_chunks, hipStream_t stream);
void launch_density(const float2 *states, float2 *d_out, int n, int batch, hipStream_t stream);
// Meyer-Wallach behind the pass that produced the state (QMLE_MEAS_MEYER_WALLACH): `last` left one
// row per tile at the start of `ws` (TM_STORE_MW / TM_MW_ONLY); d_out [batch][n + 1] = (Q, purities by wire)
bool mw_fusable(int n, const Stage &last);
// tiled state: the producing pass leaves the cross terms of positions 0..3 to the first later read (TileArgs::mw_lean)
bool mw_lean(int n, const Stage &last);
size_t mw_fused_ws_bytes(int n, int batch, const Stage &last);
int run_mw_fused(const float2 *states, int n, int batch, const Stage &last, int row_shift, void *ws,
                 size_t ws_bytes, float *d_out, hipStream_t stream);  // a row covers 2^row_shift tiles
size_t mw_resident_ws_bytes(int n, int batch);
int run_mw_resident(const float2 *states, int n, int batch, void *ws, size_t ws_bytes, float *d_out,
                    hipStream_t stream);

}  // namespace qmle
