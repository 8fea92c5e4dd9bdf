/*
 * sv_cpu.c -- C/OpenMP restatement of the reference's per-gate statevector update
 * (ORACLE / CPU BASELINE -- test infrastructure only, never linked into the product).
 *
 * Follows qml_essentials/simulation.py:65-104 (psi <- einsum(gate, psi) per gate, wire 0 =
 * most significant bit) with the contraction rule of operations.py:19-50 specialised to
 * k = 1 and k = 2:  for a gate on wires [a, b] the 4x4 row/col index is 2*bit_a + bit_b.
 * complex64 arithmetic like the reference's default (operations.py:12-16).  Checked
 * against oracle/einsum_sim.py in tests/test_oracle_c_port.py.
 *
 * Build: gcc -O3 -fcx-limited-range -fopenmp -fPIC -shared oracle/sv_cpu.c -o oracle/libsv_cpu.so
 */
#include <complex.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef float complex c64;

int svc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void svc_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

void svc_init_zero(c64 *psi, int n) {
  /* first touch in parallel with the same static partition the gate loops use, so that on a
     multi-socket host every thread's share of the state lands on its own NUMA node (a
     single-threaded memset put all pages on one node: ~31 GB/s on a 128-thread box) */
  const size_t D = (size_t)1 << n;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < D; ++i) psi[i] = 0.0f;
  psi[0] = 1.0f;
}

static inline size_t ins0(size_t i, int p) {
  return ((i >> p) << (p + 1)) | (i & (((size_t)1 << p) - 1));
}

/* m = row-major 2x2 (re,im interleaved floats); wire -> bit n-1-wire */
void svc_apply_1q(c64 *psi, int n, int wire, const float *m) {
  const int p = n - 1 - wire;
  const size_t half = (size_t)1 << (n - 1), s = (size_t)1 << p;
  const c64 m00 = m[0] + I * m[1], m01 = m[2] + I * m[3], m10 = m[4] + I * m[5],
            m11 = m[6] + I * m[7];
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < half; ++i) {
    const size_t j0 = ins0(i, p), j1 = j0 | s;
    const c64 a0 = psi[j0], a1 = psi[j1];
    psi[j0] = m00 * a0 + m01 * a1;
    psi[j1] = m10 * a0 + m11 * a1;
  }
}

/* controlled 2x2: only the control = 1 half is touched (operations.py:1074, 1397-1411) */
void svc_apply_c1q(c64 *psi, int n, int control, int target, const float *m) {
  const int pc = n - 1 - control, pt = n - 1 - target;
  const int lo = pc < pt ? pc : pt, hi = pc < pt ? pt : pc;
  const size_t quarter = (size_t)1 << (n - 2), sc = (size_t)1 << pc, st = (size_t)1 << pt;
  const c64 m00 = m[0] + I * m[1], m01 = m[2] + I * m[3], m10 = m[4] + I * m[5],
            m11 = m[6] + I * m[7];
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < quarter; ++i) {
    const size_t j0 = ins0(ins0(i, lo), hi) | sc, j1 = j0 | st;
    const c64 a0 = psi[j0], a1 = psi[j1];
    psi[j0] = m00 * a0 + m01 * a1;
    psi[j1] = m10 * a0 + m11 * a1;
  }
}

/* generic 4x4 on wires [a, b]: index = 2*bit_a + bit_b (operations.py:44-49) */
void svc_apply_2q(c64 *psi, int n, int wa, int wb, const float *m) {
  const int pa = n - 1 - wa, pb = n - 1 - wb;
  const int lo = pa < pb ? pa : pb, hi = pa < pb ? pb : pa;
  const size_t quarter = (size_t)1 << (n - 2), sa = (size_t)1 << pa, sb = (size_t)1 << pb;
  c64 M[16];
  for (int k = 0; k < 16; ++k) M[k] = m[2 * k] + I * m[2 * k + 1];
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < quarter; ++i) {
    const size_t j = ins0(ins0(i, lo), hi);
    const c64 a[4] = {psi[j], psi[j | sb], psi[j | sa], psi[j | sa | sb]};
    c64 r[4];
    for (int row = 0; row < 4; ++row)
      r[row] = M[row * 4] * a[0] + M[row * 4 + 1] * a[1] + M[row * 4 + 2] * a[2] +
               M[row * 4 + 3] * a[3];
    psi[j] = r[0];
    psi[j | sb] = r[1];
    psi[j | sa] = r[2];
    psi[j | sa | sb] = r[3];
  }
}

/* <Z_wire> for every wire: simulation.py:251-261 (one pass per observable, like the
 * reference's n_obs reductions) */
void svc_expval_z(const c64 *psi, int n, const int *wires, int n_obs, float *out) {
  const size_t D = (size_t)1 << n;
  for (int k = 0; k < n_obs; ++k) {
    const int p = n - 1 - wires[k];
    double acc = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : acc)
    for (size_t i = 0; i < D; ++i) {
      const float re = crealf(psi[i]), im = cimagf(psi[i]);
      const float pr = re * re + im * im;
      acc += ((i >> p) & 1) ? -pr : pr;
    }
    out[k] = (float)acc;
  }
}
