// libqmle_sv, complex128 engine (the reference's jax_enable_x64 mode).
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

namespace {

template <bool GMAJOR>
__global__ void __launch_bounds__(64)
k_build_matrices_f64(const BuildOp *__restrict__ build, const BuildGroup *__restrict__ groups, int n_groups,
                     const double *__restrict__ angles, int n_slots, const double *__restrict__ consts,
                     double *__restrict__ mats, uint32_t mat_floats, int batch) {
  build_matrices_body<const double *, double, double, GMAJOR>(build, groups, n_groups, angles, n_slots, consts, mats, mat_floats, batch);
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

// ---- adjoint differentiation in complex128 (qmle_adjoint_gradient_f64) ------------------------
// The complex64 sweep of qmle_adjoint.hip restated on the streaming complex128 appliers: forward state,
// lambda = (sum_k w_k Z..Z_k) psi, then per gate of the reversed, daggered tape the overlap
// Im <lambda| G |psi> of its generator and the inverse gate on psi and lambda -- one backward sweep for
// every angle where x64 mode used to contract 2 P shifted circuits (tests/test_jaqsi.py:57,131-141).
struct F64AdjTerm {
  uint32_t xmask, zmask, pmask;  // bit positions: flipped / sign / projected onto 1
  const double *marks;           // != nullptr: G = diag(marks)
};
__global__ void __launch_bounds__(256)
k64_zsum_apply(const double2 *__restrict__ psi, double2 *__restrict__ lam, int n, const double *__restrict__ weights,
               F64Obs obs, int n_obs) {
  const int b = blockIdx.y;
  const uint64_t D = (uint64_t)1 << n;
  const double *w = weights + (size_t)b * n_obs;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < D; i += (uint64_t)gridDim.x * blockDim.x) {
    double d = 0.0;
    for (int o = 0; o < n_obs; ++o) d += (__builtin_popcountll(i & obs.mask[o]) & 1) ? -w[o] : w[o];
    const double2 v = psi[((size_t)b << n) + i];
    lam[((size_t)b << n) + i] = make_double2(d * v.x, d * v.y);
  }
}
// partial[b][block] = sum_i conj(lambda_i) (X^x Z^z Pi_p psi)_i   (the phase i^n_y is applied by k64_adj_final)
__global__ void __launch_bounds__(256)
k64_adj_overlap(const double2 *__restrict__ psi_all, const double2 *__restrict__ lam_all, int n, F64AdjTerm t,
                double2 *__restrict__ partial) {
  __shared__ double red[16];
  const int b = blockIdx.y;
  const uint64_t D = (uint64_t)1 << n;
  const double2 *psi = psi_all + ((size_t)b << n), *lam = lam_all + ((size_t)b << n);
  double re = 0.0, im = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < D; i += (uint64_t)gridDim.x * blockDim.x) {
    if ((i & t.pmask) != t.pmask) continue;
    const uint64_t j = i ^ t.xmask;
    const double sg = t.marks ? t.marks[i] : ((__builtin_popcountll(j & t.zmask) & 1) ? -1.0 : 1.0);
    const double2 l = lam[i], p = psi[j];
    re += sg * (l.x * p.x + l.y * p.y);
    im += sg * (l.x * p.y - l.y * p.x);
  }
  const double r = block_sum_d(re, red);
  const double m = block_sum_d(im, red);
  if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = make_double2(r, m);
}
// grad[b][slot] = coef * Im(i^n_y * sum)
__global__ void __launch_bounds__(256)
k64_adj_final(const double2 *__restrict__ partial, int n_blocks, int n_y, double coef, double *__restrict__ grad,
              int n_grad_slots, int slot) {
  __shared__ double red[16];
  const int b = blockIdx.x;
  double re = 0.0, im = 0.0;
  for (int k = threadIdx.x; k < n_blocks; k += blockDim.x) {
    const double2 v = partial[(size_t)b * n_blocks + k];
    re += v.x;
    im += v.y;
  }
  const double r = block_sum_d(re, red);
  const double i = block_sum_d(im, red);
  if (threadIdx.x == 0) {
    const int q = n_y & 3;
    grad[(size_t)b * n_grad_slots + slot] = coef * (q == 0 ? i : q == 1 ? r : q == 2 ? -i : -r);
  }
}

}  // namespace

// ---- complex128 engine: host side ------------------------------------------------------------
static int ensure_f64(qmle_plan *p) {
  if (p->f64_blob) return p->f64_device == current_device() ? QMLE_OK : QMLE_ERR_UNSUPPORTED;
  const size_t b_low = align_up(p->lowered.size() * sizeof(LoweredOp) + 16, 256);
  const size_t nc = p->n_user_consts;
  const size_t b_c = align_up(nc * sizeof(double) + 16, 256);
  char *blob = nullptr;
  HIPCHK(hipMalloc((void **)&blob, b_low + b_c));
  // (the plan takes the blob only once BOTH uploads have succeeded: a failed copy must not leave a
  // plan that looks initialised and runs on garbage operators)
  hipError_t err = hipSuccess;
  if (!p->lowered.empty())
    err = hipMemcpy(blob, p->lowered.data(), p->lowered.size() * sizeof(LoweredOp), hipMemcpyHostToDevice);
  if (err == hipSuccess && nc) {
    std::vector<double> c64(nc);
    for (size_t i = 0; i < nc; ++i) c64[i] = i < p->consts64.size() ? p->consts64[i] : (double)p->consts[i];
    err = hipMemcpy(blob + b_low, c64.data(), nc * sizeof(double), hipMemcpyHostToDevice);
  }
  if (err != hipSuccess) {
    (void)hipFree(blob);
    g_last_hip_error = (int)err;
    return QMLE_ERR_HIP;
  }
  p->f64_blob = blob;
  p->f64_device = current_device();
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
    if (batch >= 64)
      hipLaunchKernelGGL(k_build_matrices_f64<true>, dim3(grid_for((uint64_t)ng * (((uint64_t)batch + 63) / 64) * 64, 64)), dim3(64), 0, stream, plan->dev.d_build,
                       plan->dev.d_groups, ng, d_angles, plan->n_slots, d_c64, d_mats, plan->mat_floats, batch);
    else
      hipLaunchKernelGGL(k_build_matrices_f64<false>, dim3(grid_for((uint64_t)batch * ng, 64)), dim3(64), 0, stream, plan->dev.d_build,
                       plan->dev.d_groups, ng, d_angles, plan->n_slots, d_c64, d_mats, plan->mat_floats, batch);
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
    if (batch >= 64)
      hipLaunchKernelGGL(k_build_matrices_f64<true>, dim3(grid_for((uint64_t)ng * (((uint64_t)batch + 63) / 64) * 64, 64)), dim3(64), 0, stream, plan->dev.d_build,
                       plan->dev.d_groups, ng, d_angles, plan->n_slots, d_c64, d_mats, plan->mat_floats, batch);
    else
      hipLaunchKernelGGL(k_build_matrices_f64<false>, dim3(grid_for((uint64_t)batch * ng, 64)), dim3(64), 0, stream, plan->dev.d_build,
                       plan->dev.d_groups, ng, d_angles, plan->n_slots, d_c64, d_mats, plan->mat_floats, batch);
  }
  const size_t D = (size_t)1 << n;
  const int n_ops = (int)plan->lowered.size();
  double2 *d_states = (double2 *)ws;
  if (n <= 13) {
    if (FirstUse once{6}; once.first) {
      HIPCHK(hipFuncSetAttribute((const void *)k64_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
      once.done();
    }
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

// ---- adjoint gradient, complex128 --------------------------------------------------------------
static int f64_adj_blocks(int n) {
  const uint64_t D = (uint64_t)1 << n;
  uint64_t b = (D + 256 * 8 - 1) / (256 * 8);
  return (int)(b < 1 ? 1 : b > 1024 ? 1024 : b);
}
static int f64_adj_in_flight(int n, int batch) {  // psi and lambda of a round of samples: <= 4 GiB
  size_t s = ((size_t)4 << 30) / ((size_t)32 << n);
  if (s < 1) s = 1;
  if (s > (size_t)batch) s = (size_t)batch;
  if (s > 65535) s = 65535;
  return (int)s;
}
struct F64AdjLayout { size_t fmats, rmats, partial, psi, lam, total; };
static F64AdjLayout f64_adj_layout(const qmle_plan *fwd, const qmle_plan *rev, int batch) {
  F64AdjLayout L;
  const int fl = f64_adj_in_flight(fwd->n, batch);
  L.fmats = 0;
  L.rmats = align_up((size_t)batch * (fwd->mat_floats ? fwd->mat_floats : 1) * sizeof(double), 256);
  L.partial = L.rmats + align_up((size_t)batch * (rev->mat_floats ? rev->mat_floats : 1) * sizeof(double), 256);
  L.psi = L.partial + align_up((size_t)fl * f64_adj_blocks(fwd->n) * sizeof(double2), 256);
  L.lam = L.psi + align_up((size_t)fl * ((size_t)16 << fwd->n), 256);
  L.total = L.lam + align_up((size_t)fl * ((size_t)16 << fwd->n), 256) + 512;
  return L;
}

size_t qmle_adjoint_workspace_bytes_f64(const qmle_plan *fwd, const qmle_plan *rev, int batch) {
  if (!fwd || !rev || batch < 1 || fwd->n != rev->n) return 0;
  return f64_adj_layout(fwd, rev, batch).total;
}

int qmle_adjoint_gradient_f64(qmle_plan *fwd, qmle_plan *rev, const double *d_angles_fwd, const double *d_angles_rev,
                              int batch, const double *d_weights, const uint32_t *obs_wire_masks, int n_obs,
                              const qmle_adjoint_term *terms, int n_terms, double *d_grad, int n_grad_slots,
                              void *d_workspace, size_t workspace_bytes, qmle_stream stream_) {
  if (!fwd || !rev || batch < 1 || !d_weights || !obs_wire_masks || n_obs < 1 || n_obs > QMLE_MAX_QUBITS || !terms ||
      !d_grad || n_grad_slots < 1 || !d_workspace || fwd->n != rev->n || n_terms != (int)rev->ops.size())
    return QMLE_ERR_INVALID_ARG;
  if ((fwd->n_slots > 0 && !d_angles_fwd) || (rev->n_slots > 0 && !d_angles_rev)) return QMLE_ERR_INVALID_ARG;
  const int n = fwd->n;
  // one source gate (one generator) per operator of the reverse plan: NO_FUSION or NO_MERGE plans
  for (const auto &srcs : rev->lowered_src)
    if (srcs.size() != 1) return QMLE_ERR_INVALID_ARG;
  F64Obs obs;
  std::memset(&obs, 0, sizeof(obs));
  for (int k = 0; k < n_obs; ++k) {
    const uint32_t in = obs_wire_masks[k];
    if (in == 0 || (n < 32 && (in >> n))) return QMLE_ERR_WIRE_RANGE;
    for (int w = 0; w < n; ++w)
      if (in & (1u << w)) obs.mask[k] |= 1u << (n - 1 - w);
  }
  for (int r = 0; r < n_terms; ++r) {
    if (terms[r].out_slot >= n_grad_slots) return QMLE_ERR_SLOT_RANGE;
    if (terms[r].out_slot >= 0 && terms[r].marks_off >= 0 &&
        (size_t)terms[r].marks_off + ((size_t)1 << n) > rev->n_user_consts)
      return QMLE_ERR_INVALID_ARG;
  }
  const F64AdjLayout L = f64_adj_layout(fwd, rev, batch);
  char *ws = (char *)d_workspace;
  const size_t mis = (size_t)(256 - ((uintptr_t)ws & 255)) & 255;
  if (workspace_bytes < mis + L.total) return QMLE_ERR_WORKSPACE;
  ws += mis;
  hipStream_t stream = (hipStream_t)stream_;
  int rc = ensure_device_plan(fwd);
  if (rc == QMLE_OK) rc = ensure_device_plan(rev);
  if (rc == QMLE_OK) rc = ensure_f64(fwd);
  if (rc == QMLE_OK) rc = ensure_f64(rev);
  if (rc != QMLE_OK) return rc;
  auto c64_of = [](const qmle_plan *p) {
    return (const double *)((char *)p->f64_blob + align_up(p->lowered.size() * sizeof(LoweredOp) + 16, 256));
  };
  const double *fc = c64_of(fwd), *rcst = c64_of(rev);
  double *fmats = (double *)(ws + L.fmats), *rmats = (double *)(ws + L.rmats);
  auto build = [&](const qmle_plan *p, const double *ang, const double *c64, double *mats) {
    if (p->groups.empty()) return;
    const int ng = (int)p->groups.size();
    if (batch >= 64)
      hipLaunchKernelGGL(k_build_matrices_f64<true>, dim3(grid_for((uint64_t)ng * (((uint64_t)batch + 63) / 64) * 64, 64)),
                         dim3(64), 0, stream, p->dev.d_build, p->dev.d_groups, ng, ang, p->n_slots, c64, mats,
                         p->mat_floats, batch);
    else
      hipLaunchKernelGGL(k_build_matrices_f64<false>, dim3(grid_for((uint64_t)batch * ng, 64)), dim3(64), 0, stream,
                         p->dev.d_build, p->dev.d_groups, ng, ang, p->n_slots, c64, mats, p->mat_floats, batch);
  };
  build(fwd, d_angles_fwd, fc, fmats);
  build(rev, d_angles_rev, rcst, rmats);
  HIPCHK(hipMemsetAsync(d_grad, 0, (size_t)batch * n_grad_slots * sizeof(double), stream));
  double2 *psi = (double2 *)(ws + L.psi), *lam = (double2 *)(ws + L.lam);
  double2 *partial = (double2 *)(ws + L.partial);
  const size_t D = (size_t)1 << n;
  const int nb = f64_adj_blocks(n);
  const int in_flight = f64_adj_in_flight(n, batch);
  const unsigned gx = grid_for(D / 2 ? D / 2 : 1, 256, 1u << 16);
  for (int b0 = 0; b0 < batch; b0 += in_flight) {
    const int bc = batch - b0 < in_flight ? batch - b0 : in_flight;
    const double *af = d_angles_fwd ? d_angles_fwd + (size_t)b0 * fwd->n_slots : nullptr;
    const double *ar = d_angles_rev ? d_angles_rev + (size_t)b0 * rev->n_slots : nullptr;
    // forward: psi = U_N .. U_1 |0>
    hipLaunchKernelGGL(k64_init, dim3(gx, bc), dim3(256), 0, stream, psi, n);
    for (const LoweredOp &op : fwd->lowered)
      hipLaunchKernelGGL(k64_op, dim3(gx, bc), dim3(256), 0, stream, psi, n, op, fmats + (size_t)b0 * fwd->mat_floats,
                         fwd->mat_floats, fc, af, fwd->n_slots);
    // lambda = (sum_k w_k Z..Z_k) psi
    hipLaunchKernelGGL(k64_zsum_apply, dim3(gx, bc), dim3(256), 0, stream, (const double2 *)psi, lam, n,
                       d_weights + (size_t)b0 * n_obs, obs, n_obs);
    for (size_t r = 0; r < rev->lowered.size(); ++r) {
      const qmle_adjoint_term &t = terms[rev->lowered_src[r][0]];
      if (t.out_slot >= 0) {
        F64AdjTerm a;
        a.xmask = a.zmask = a.pmask = 0;
        for (int w = 0; w < n; ++w) {
          if (t.x_wires & (1u << w)) a.xmask |= 1u << (n - 1 - w);
          if (t.z_wires & (1u << w)) a.zmask |= 1u << (n - 1 - w);
          if (t.proj_wires & (1u << w)) a.pmask |= 1u << (n - 1 - w);
        }
        a.marks = t.marks_off >= 0 ? rcst + t.marks_off : nullptr;
        hipLaunchKernelGGL(k64_adj_overlap, dim3(nb, bc), dim3(256), 0, stream, (const double2 *)psi,
                           (const double2 *)lam, n, a, partial);
        hipLaunchKernelGGL(k64_adj_final, dim3(bc), dim3(nb >= 256 ? 256 : 64), 0, stream, (const double2 *)partial, nb,
                           t.n_y, (double)t.coef, d_grad + (size_t)b0 * n_grad_slots, n_grad_slots, t.out_slot);
      }
      for (double2 *st : {psi, lam})
        hipLaunchKernelGGL(k64_op, dim3(gx, bc), dim3(256), 0, stream, st, n, rev->lowered[r],
                           rmats + (size_t)b0 * rev->mat_floats, rev->mat_floats, rcst, ar, rev->n_slots);
    }
    HIPCHK(hipGetLastError());
  }
  return QMLE_OK;
}

}  // extern "C"
