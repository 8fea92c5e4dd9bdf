// libqmle_sv, measurement and analysis kernels of resident states: <Z> / parities, probabilities,
// density matrices, marginals, overlaps and pair fidelities, Meyer-Wallach, histogram, the parameter
// sampler and shot sampling -- with their C-ABI entry points.
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

namespace {

// ---------------------------------------------------------------------------
// all-qubit <Z> in one read.  Element index bits: bit0 = position inside the
// float4 chunk, bits 1..8 = thread id, bits 9..10 = unroll slot u, bits >= 11 =
// segment id.  Every bit gets its own signed accumulator (static indexing).
// ---------------------------------------------------------------------------
constexpr int kEzThreads = 256;
constexpr int kEzUnroll = 4;
constexpr int kEzSegBits = 11;  // 1 + 8 + 2
constexpr int kEzMaxHigh = QMLE_MAX_QUBITS - kEzSegBits;

__global__ void __launch_bounds__(kEzThreads)
k_expval_partial(const float4 *__restrict__ states, int n, float *__restrict__ partial) {
  __shared__ float red[16];
  const int b = blockIdx.y, tid = threadIdx.x;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  const float4 *st = states + (size_t)b * chunks;
  const uint64_t seg_chunks = (uint64_t)kEzThreads * kEzUnroll;
  const uint64_t n_seg = (chunks + seg_chunks - 1) / seg_chunks;
  float acc_tot = 0.f, acc_b0 = 0.f, acc_u0 = 0.f, acc_u1 = 0.f;
  float acc_hi[kEzMaxHigh];
#pragma unroll
  for (int k = 0; k < kEzMaxHigh; ++k) acc_hi[k] = 0.f;
  for (uint64_t seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
    float4 v[kEzUnroll];
#pragma unroll
    for (int u = 0; u < kEzUnroll; ++u) {
      const uint64_t c = seg * seg_chunks + (uint64_t)u * kEzThreads + tid;
      v[u] = c < chunks ? st[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float pe = 0.f, po = 0.f, pu[kEzUnroll];
#pragma unroll
    for (int u = 0; u < kEzUnroll; ++u) {
      const float e = v[u].x * v[u].x + v[u].y * v[u].y;
      const float o = v[u].z * v[u].z + v[u].w * v[u].w;
      pe += e;
      po += o;
      pu[u] = e + o;
    }
    const float tot = pe + po;
    acc_tot += tot;
    acc_b0 += pe - po;
    acc_u0 += (pu[0] - pu[1]) + (pu[2] - pu[3]);
    acc_u1 += (pu[0] + pu[1]) - (pu[2] + pu[3]);
#pragma unroll
    for (int k = 0; k < kEzMaxHigh; ++k) acc_hi[k] += ((seg >> k) & 1ull) ? -tot : tot;
  }
  // partial[b][block][bit]; bit n is the plain total (norm check)
  float *out = partial + ((size_t)b * gridDim.x + blockIdx.x) * (QMLE_MAX_QUBITS + 1);
  float r;
  r = block_sum(acc_b0, red);
  if (tid == 0) out[0] = r;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    r = block_sum(((tid >> i) & 1) ? -acc_tot : acc_tot, red);
    if (tid == 0) out[1 + i] = r;
  }
  r = block_sum(acc_u0, red);
  if (tid == 0) out[9] = r;
  r = block_sum(acc_u1, red);
  if (tid == 0) out[10] = r;
#pragma unroll
  for (int k = 0; k < kEzMaxHigh; ++k) {
    r = block_sum(acc_hi[k], red);
    if (tid == 0) out[kEzSegBits + k] = r;
  }
  r = block_sum(acc_tot, red);
  if (tid == 0) out[QMLE_MAX_QUBITS] = r;
}


// one block per (state, observable): fp64 sum of that bit's column over all partial rows
__global__ void __launch_bounds__(256)
k_expval_final(const float *__restrict__ partial, int n_blocks, int n_obs, ObsBits obs,
               float *__restrict__ out) {
  __shared__ double red[16];
  const int b = blockIdx.x, k = blockIdx.y;
  const float *pp = partial + (size_t)b * n_blocks * (QMLE_MAX_QUBITS + 1) + obs.bits[k];
  const uint32_t rm = obs.row_mask[k];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_blocks; i += blockDim.x) {
    const double v = (double)pp[(size_t)i * (QMLE_MAX_QUBITS + 1)];
    acc += (__popc(rm & (uint32_t)i) & 1) ? -v : v;
  }
  const double tot = block_sum_d(acc, red);
  if (threadIdx.x == 0) out[(size_t)b * n_obs + k] = (float)tot;
}

// ---------------------------------------------------------------------------
// simple streaming kernels
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_probs(const float4 *__restrict__ states, float2 *__restrict__ out, uint64_t total_chunks) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total_chunks; k += stride) {
    const float4 v = states[k];
    out[k] = make_float2(v.x * v.x + v.y * v.y, v.z * v.z + v.w * v.w);
  }
}

__global__ void __launch_bounds__(256)
k_density(const float2 *__restrict__ states, float2 *__restrict__ out, int n) {
  const int b = blockIdx.y;
  const uint64_t D = (uint64_t)1 << n;
  const float2 *st = states + (size_t)b * D;
  float2 *o = out + (size_t)b * D * D;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < D * D; k += stride) {
    const uint64_t i = k >> n, j = k & (D - 1);
    const float2 a = st[i], c = st[j];
    o[k] = make_float2(a.x * c.x + a.y * c.y, a.y * c.x - a.x * c.y);  // a * conj(c)
  }
}

struct KeepBits {
  int8_t bits[QMLE_MAX_QUBITS];  // bit positions of kept wires, LSB of output first
  int n_keep;
};

// Few kept wires (<= 12): every workgroup bins its share of |a|^2 in LDS first and adds one value
// per bin to the output -- 2^n / grid terms per global atomic instead of one.  (One global float
// atomic per amplitude left partial probabilities 2e-6 .. 2e-5 apart between two runs on the
// same state at n = 18, and it is the slowest way to add.)
__global__ void __launch_bounds__(256)
k_marginal_lds(const float2 *__restrict__ states, float *__restrict__ out, int n, KeepBits kb) {
  extern __shared__ float4 smem4[];
  float *bins = reinterpret_cast<float *>(smem4);
  const int b = blockIdx.y;
  const uint32_t n_bins = 1u << kb.n_keep;
  for (uint32_t k = threadIdx.x; k < n_bins; k += blockDim.x) bins[k] = 0.f;
  __syncthreads();
  const uint64_t D = (uint64_t)1 << n;
  const float2 *st = states + (size_t)b * D;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < D; i += stride) {
    uint32_t idx = 0;
    for (int k = 0; k < kb.n_keep; ++k) idx |= (uint32_t)((i >> kb.bits[k]) & 1ull) << k;
    atomicAdd(bins + idx, norm2(st[i]));
  }
  __syncthreads();
  float *o = out + ((size_t)b << kb.n_keep);
  for (uint32_t k = threadIdx.x; k < n_bins; k += blockDim.x) {
    const float v = bins[k];
    if (v != 0.f) atomicAdd(o + k, v);
  }
}

__global__ void __launch_bounds__(256)
k_marginal(const float2 *__restrict__ states, float *__restrict__ out, int n, KeepBits kb) {
  const int b = blockIdx.y;
  const uint64_t D = (uint64_t)1 << n;
  const float2 *st = states + (size_t)b * D;
  float *o = out + ((size_t)b << kb.n_keep);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < D; i += stride) {
    uint32_t idx = 0;
    for (int k = 0; k < kb.n_keep; ++k) idx |= (uint32_t)((i >> kb.bits[k]) & 1ull) << k;
    atomicAdd(o + idx, norm2(st[i]));
  }
}

// <a|b> partial sums: partial[pair][block] = (re, im)
__global__ void __launch_bounds__(256)
k_overlap_partial(const float4 *__restrict__ states, int n, int n_pairs,
                  float2 *__restrict__ partial) {
  __shared__ float red[16];
  const int pr = blockIdx.y;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  const float4 *a = states + (size_t)pr * chunks;
  const float4 *c = states + ((size_t)pr + n_pairs) * chunks;
  float re = 0.f, im = 0.f;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < chunks; k += stride) {
    const float4 x = a[k], y = c[k];
    // conj(x) * y
    re += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    im += x.x * y.y - x.y * y.x + x.z * y.w - x.w * y.z;
  }
  const float r = block_sum(re, red);
  const float i = block_sum(im, red);
  if (threadIdx.x == 0) partial[(size_t)pr * gridDim.x + blockIdx.x] = make_float2(r, i);
}

__global__ void __launch_bounds__(256)
k_overlap_final(const float2 *__restrict__ partial, int n_blocks, int n_pairs,
                float *__restrict__ out) {
  __shared__ double red[16];
  const int pr = blockIdx.x;
  double re = 0.0, im = 0.0;
  for (int i = threadIdx.x; i < n_blocks; i += blockDim.x) {
    const float2 v = partial[(size_t)pr * n_blocks + i];
    re += v.x;
    im += v.y;
  }
  re = block_sum_d(re, red);
  im = block_sum_d(im, red);
  if (threadIdx.x == 0) out[pr] = (float)(re * re + im * im);
}
// Small states (n <= 16): ONE workgroup per pair, no partial rows -- the two-launch form spent 17 us
// on 1024 pairs of 12-qubit states, most of it launch latency and the row round trip.  fp32 inside a
// thread (<= 128 products each), fp64 across the workgroup.
__global__ void __launch_bounds__(256)
k_pair_fidelity_small(const float4 *__restrict__ states, int n, int n_pairs, float *__restrict__ out) {
  __shared__ double red[16];
  const int pr = blockIdx.x;
  const uint32_t chunks = 1u << (n - 1);
  const float4 *a = states + (size_t)pr * chunks;
  const float4 *c = states + ((size_t)pr + n_pairs) * chunks;
  float re = 0.f, im = 0.f;
  for (uint32_t k = threadIdx.x; k < chunks; k += blockDim.x) {
    const float4 x = a[k], y = c[k];  // conj(x) * y
    re += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    im += x.x * y.y - x.y * y.x + x.z * y.w - x.w * y.z;
  }
  const double r = block_sum_d((double)re, red);
  const double i = block_sum_d((double)im, red);
  if (threadIdx.x == 0) out[pr] = (float)(r * r + i * i);
}

// Meyer-Wallach cross terms c_j = sum_{bit_j = 0} psi_i conj(psi_{i + 2^j}) plus the
// populations a_j, d_j; one launch per bit (v1: n reads of the state).
// ---- parameter sampler on the device ---------------------------------------------------------
// numpy's Philox4x64-10 stream (csrc/qmle_rng.cpp restates it on the host): block b of the stream
// is philox(counter = b + 1, key) -- any block on its own, one work item per block of four values.
// The arithmetic after the generator is numpy's, rounding for rounding: u = (x >> 11) * 2^-53
// (exact), low + range * u as a rounded product and a rounded sum (no fused multiply-add), cast to
// float32.  Expressibility(12 q, 1024 pairs) spent 0.4 of its 0.9 ms drawing parameters on the host.
__device__ __forceinline__ void philox_uniform_body(uint64_t k0, uint64_t k1, uint64_t n, double low, double range,
                                                    float *__restrict__ out) {
  const uint64_t blocks = (n + 3) / 4;
  for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < blocks; b += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t c0 = b + 1, c1 = 0, c2 = 0, c3 = 0, a0 = k0, a1 = k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const uint64_t M0 = 0xD2E7470EE14C6C93ull, M1 = 0xCA5A826395121157ull;
      const uint64_t hi0 = __umul64hi(M0, c0), lo0 = M0 * c0, hi1 = __umul64hi(M1, c2), lo1 = M1 * c2;
      const uint64_t n0 = hi1 ^ c1 ^ a0, n2 = hi0 ^ c3 ^ a1;
      c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
      a0 += 0x9E3779B97F4A7C15ull;
      a1 += 0xBB67AE8584CAA73Bull;
    }
    const uint64_t v[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t i = 4 * b + (uint64_t)j;
      if (i < n) {
        const double u = __dmul_rn((double)(v[j] >> 11), 1.0 / 9007199254740992.0);
        out[i] = (float)__dadd_rn(low, __dmul_rn(range, u));
      }
    }
  }
}
__global__ void __launch_bounds__(256)
k_philox_uniform(uint64_t k0, uint64_t k1, uint64_t n, double low, double range, float *__restrict__ out) {
  philox_uniform_body(k0, k1, n, low, range, out);
}
// the key read from device memory: a captured launch (hipGraph) is re-seeded by rewriting two words
__global__ void __launch_bounds__(256)
k_philox_uniform_devkey(const uint64_t *__restrict__ key, uint64_t n, double low, double range, float *__restrict__ out) {
  philox_uniform_body(key[0], key[1], n, low, range, out);
}
// partial[b][bit][block] = (re c, im c, a, d)
__global__ void __launch_bounds__(256)
k_cross_partial(const float4 *__restrict__ states, int n, int p, float4 *__restrict__ partial,
                int n_blocks) {
  __shared__ float red[16];
  const int b = blockIdx.y;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  const float4 *st = states + (size_t)b * chunks;
  float cr = 0.f, ci = 0.f, pa = 0.f, pd = 0.f;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  if (p == 0) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < chunks; k += stride) {
      const float4 v = st[k];
      cr += v.x * v.z + v.y * v.w;
      ci += v.y * v.z - v.x * v.w;
      pa += v.x * v.x + v.y * v.y;
      pd += v.z * v.z + v.w * v.w;
    }
  } else {
    const uint64_t items = chunks >> 1;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < items; k += stride) {
      const uint64_t c0 = ins0_64(k, p - 1), c1 = c0 | (1ull << (p - 1));
      const float4 x = st[c0], y = st[c1];
      cr += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      ci += x.y * y.x - x.x * y.y + x.w * y.z - x.z * y.w;
      pa += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
      pd += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
    }
  }
  const float r0 = block_sum(cr, red), r1 = block_sum(ci, red), r2 = block_sum(pa, red),
              r3 = block_sum(pd, red);
  if (threadIdx.x == 0)
    partial[((size_t)b * n + p) * n_blocks + blockIdx.x] = make_float4(r0, r1, r2, r3);
}

__global__ void __launch_bounds__(256)
k_mw_final(const float4 *__restrict__ partial, int n, int n_blocks, int batch,
           float *__restrict__ out, float *__restrict__ purities) {
  __shared__ double red[16];
  const int b = blockIdx.x;
  double sum = 0.0;
  for (int p = 0; p < n; ++p) {
    double cr = 0, ci = 0, a = 0, d = 0;
    for (int i = threadIdx.x; i < n_blocks; i += blockDim.x) {
      const float4 v = partial[((size_t)b * n + p) * n_blocks + i];
      cr += v.x; ci += v.y; a += v.z; d += v.w;
    }
    cr = block_sum_d(cr, red);
    ci = block_sum_d(ci, red);
    a = block_sum_d(a, red);
    d = block_sum_d(d, red);
    if (threadIdx.x == 0) {
      const double pur = a * a + d * d + 2.0 * (cr * cr + ci * ci);
      if (purities) purities[(size_t)b * n + (n - 1 - p)] = (float)pur;  // index by wire
      sum += pur;
    }
  }
  if (threadIdx.x == 0) out[b] = (float)(2.0 * (1.0 - sum / n));
}

// numpy.histogram's bin of v over linspace(lo, hi, n_bins + 1), last bin right-inclusive; -1 = dropped
__device__ __forceinline__ int hist_bin(float v, int n_bins, float lo, float hi) {
  if (!(v >= lo) || !(v <= hi)) return -1;  // numpy drops out-of-range and NaN
  const float scale = (float)n_bins / (hi - lo);
  int bin = (int)((v - lo) * scale);
  if (bin >= n_bins) bin = n_bins - 1;     // right edge inclusive
  // guard against rounding across an edge: edges are lo + k*(hi-lo)/n_bins
  const float w = (hi - lo) / (float)n_bins;
  if (bin > 0 && v < lo + bin * w) --bin;
  else if (bin < n_bins - 1 && v >= lo + (bin + 1) * w) ++bin;
  return bin;
}

__global__ void __launch_bounds__(256)
k_histogram(const float *__restrict__ values, int64_t count, int n_bins, float lo, float hi,
            int *__restrict__ counts) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    const int bin = hist_bin(values[i], n_bins, lo, hi);
    if (bin >= 0) atomicAdd(counts + bin, 1);
  }
}

// <= 4096 bins: every workgroup bins its share in LDS (integer atomics on the LDS crossbar instead
// of contended global ones: 1024 values into 75 bins took 13.8 us + a 5 us memset before).  ONE
// workgroup (SOLO, counts up to 2^16 values): the bins are stored, not added -- no memset launch.
template <bool SOLO>
__global__ void __launch_bounds__(1024)
k_histogram_lds(const float *__restrict__ values, int64_t count, int n_bins, float lo, float hi,
                int *__restrict__ counts) {
  extern __shared__ float4 smem4[];
  int *bins = reinterpret_cast<int *>(smem4);
  for (int k = threadIdx.x; k < n_bins; k += blockDim.x) bins[k] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    const int bin = hist_bin(values[i], n_bins, lo, hi);
    if (bin >= 0) atomicAdd(bins + bin, 1);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < n_bins; k += blockDim.x) {
    if (SOLO) counts[k] = bins[k];
    else if (bins[k]) atomicAdd(counts + k, bins[k]);
  }
}


// ---- shot sampling (simulation.py:320-377) ------------------------------------------
// Philox4x32-10 counter RNG: counter = (shot pair, 0, row lo, row hi), key = seed.
struct Philox4 { uint32_t x[4]; };
__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                 uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{{c0, c1, c2, c3}};
}
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {  // uniform in (0, 1)
  return ((double)((((uint64_t)hi << 32) | lo) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

// Inclusive fp64 prefix sum of one row of probabilities per block: cdf[b][i] = sum_{j<=i} p[b][j]
__global__ void __launch_bounds__(256)
k_cdf(const float *__restrict__ probs, uint64_t D, double *__restrict__ cdf) {
  __shared__ double wsum[4];
  __shared__ double carry_s;
  const float *p = probs + (size_t)blockIdx.x * D;
  double *c = cdf + (size_t)blockIdx.x * D;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0.0;
  __syncthreads();
  for (uint64_t base = 0; base < D; base += 1024) {
    const uint64_t i0 = base + (uint64_t)threadIdx.x * 4;
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (i0 + k < D) ? (double)p[i0 + k] : 0.0;
    v[1] += v[0];
    v[2] += v[1];
    v[3] += v[2];
    double incl = v[3];  // inclusive scan of the per-thread totals across the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    double before = carry_s + (incl - v[3]);
    for (int j = 0; j < w; ++j) before += wsum[j];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i0 + k < D) c[i0 + k] = before + v[k];
    __syncthreads();
    if (threadIdx.x == 255) carry_s = before + v[3];
    __syncthreads();
  }
}

// First index with cdf[idx] >= r (numpy.searchsorted side="left").
__device__ __forceinline__ uint32_t cdf_search(const double *__restrict__ c, uint64_t D,
                                               double r) {
  uint64_t lo = 0, hi = D - 1;  // answer in [lo, hi]; cdf[D-1] = total >= r
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (c[mid] >= r) hi = mid; else lo = mid + 1;
  }
  return (uint32_t)lo;
}

constexpr int kShotsPerThread = 16;                       // 8 Philox blocks
constexpr int kShotsPerBlock = 256 * kShotsPerThread;     // 4096
constexpr int kLdsHistMax = 4096;                         // bins kept in LDS (16 KiB)

// counts[b][idx] += 1 for `shots` draws idx ~ probs[b]; grid (ceil(shots/4096), rows)
template <bool LDS_HIST>
__global__ void __launch_bounds__(256)
k_sample(const double *__restrict__ cdf, uint64_t D, int shots, uint64_t seed,
         uint64_t row_offset, int *__restrict__ counts) {
  __shared__ int hist[LDS_HIST ? kLdsHistMax : 1];
  const double *c = cdf + (size_t)blockIdx.y * D;
  int *out = counts + (size_t)blockIdx.y * D;
  if (LDS_HIST) {
    for (uint32_t i = threadIdx.x; i < D; i += 256) hist[i] = 0;
    __syncthreads();
  }
  const double total = c[D - 1];
  const uint64_t row = row_offset + blockIdx.y;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const int64_t first = (int64_t)blockIdx.x * kShotsPerBlock;
#pragma unroll 1
  for (int j = 0; j < kShotsPerThread / 2; ++j) {
    // shot pair index: consecutive threads take consecutive pairs
    const int64_t pair = first / 2 + (int64_t)j * 256 + threadIdx.x;
    if (2 * pair >= shots) break;
    const Philox4 rnd = philox4x32_10((uint32_t)pair, (uint32_t)((uint64_t)pair >> 32),
                                      (uint32_t)row, (uint32_t)(row >> 32), k0, k1);
    const uint32_t a = cdf_search(c, D, total * u53(rnd.x[0], rnd.x[1]));
    if (LDS_HIST) atomicAdd(&hist[a], 1); else atomicAdd(out + a, 1);
    if (2 * pair + 1 < shots) {
      const uint32_t b = cdf_search(c, D, total * u53(rnd.x[2], rnd.x[3]));
      if (LDS_HIST) atomicAdd(&hist[b], 1); else atomicAdd(out + b, 1);
    }
  }
  if (LDS_HIST) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < D; i += 256)
      if (hist[i]) atomicAdd(out + i, hist[i]);
  }
}

__global__ void __launch_bounds__(256)
k_counts_to_probs(const int *__restrict__ counts, uint64_t total, float inv_shots,
                  float *__restrict__ out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
    out[i] = (float)counts[i] * inv_shots;
}

// sum_i p[b][i] * d_k[sub_k(i)]: the diagonal of observable k lifted to the register
// (simulation.py:363-372).  diag_off < 0: Z-parity over the wires (no table).
struct DiagObs {
  int8_t bits[QMLE_MAX_QUBITS];  // bit positions of the observable's wires, MSB first
  int n_wires;
  int diag_off;
};
__global__ void __launch_bounds__(256)
k_probs_diag_expval(const float *__restrict__ probs, uint64_t D, const DiagObs *__restrict__ obs,
                    const float *__restrict__ diag, int n_obs, float *__restrict__ out) {
  __shared__ double red[16];
  const DiagObs ob = obs[blockIdx.y];
  const float *p = probs + (size_t)blockIdx.x * D;
  double acc = 0.0;
  for (uint64_t i = threadIdx.x; i < D; i += 256) {
    uint32_t sub = 0;
    for (int k = 0; k < ob.n_wires; ++k) sub = (sub << 1) | (uint32_t)((i >> ob.bits[k]) & 1);
    const float d = ob.diag_off < 0 ? ((__popc(sub) & 1) ? -1.f : 1.f) : diag[ob.diag_off + sub];
    acc += (double)(p[i] * d);
  }
  const double t = block_sum_d(acc, red);
  if (threadIdx.x == 0) out[(size_t)blockIdx.x * n_obs + blockIdx.y] = (float)t;
}



// <a_i|b_i> for separate arrays a, b: partial[i][block] = (re, im)
__global__ void __launch_bounds__(256)
k_overlap2_partial(const float4 *__restrict__ a_all, const float4 *__restrict__ b_all, int n,
                   float2 *__restrict__ partial) {
  __shared__ float red[16];
  const int pr = blockIdx.y;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  const float4 *a = a_all + (size_t)pr * chunks;
  const float4 *c = b_all + (size_t)pr * chunks;
  float re = 0.f, im = 0.f;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < chunks; k += stride) {
    const float4 x = a[k], y = c[k];
    re += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    im += x.x * y.y - x.y * y.x + x.z * y.w - x.w * y.z;
  }
  const float r = block_sum(re, red);
  const float i = block_sum(im, red);
  if (threadIdx.x == 0) partial[(size_t)pr * gridDim.x + blockIdx.x] = make_float2(r, i);
}

__global__ void __launch_bounds__(256)
k_overlap2_final(const float2 *__restrict__ partial, int n_blocks, int count,
                 float2 *__restrict__ out) {
  __shared__ double red[16];
  const int pr = blockIdx.x;
  double re = 0.0, im = 0.0;
  for (int i = threadIdx.x; i < n_blocks; i += blockDim.x) {
    const float2 v = partial[(size_t)pr * n_blocks + i];
    re += v.x;
    im += v.y;
  }
  re = block_sum_d(re, red);
  im = block_sum_d(im, red);
  if (threadIdx.x == 0) out[pr] = make_float2((float)re, (float)im);
}

// Z-parity expectation: sum_i (-1)^{popcount(i & mask)} |psi_i|^2, up to 8 masks per launch.
struct ParityMasks {
  uint32_t m[8];
  int count;
};

__global__ void __launch_bounds__(256)
k_parity_partial(const float4 *__restrict__ states, int n, ParityMasks pm,
                 float *__restrict__ partial) {
  __shared__ float red[16];
  const int b = blockIdx.y;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  const float4 *st = states + (size_t)b * chunks;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += stride) {
    const float4 v = st[c];
    const float pe = v.x * v.x + v.y * v.y, po = v.z * v.z + v.w * v.w;
    const uint32_t ie = (uint32_t)(c << 1), io = ie | 1u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc[k] += (__popc(ie & pm.m[k]) & 1) ? -pe : pe;
      acc[k] += (__popc(io & pm.m[k]) & 1) ? -po : po;
    }
  }
  float *out = partial + ((size_t)b * gridDim.x + blockIdx.x) * 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float r = block_sum(acc[k], red);
    if (threadIdx.x == 0) out[k] = r;
  }
}

__global__ void __launch_bounds__(256)
k_parity_final(const float *__restrict__ partial, int n_blocks, int count, int n_obs_total,
               int obs_off, float *__restrict__ out) {
  __shared__ double red[16];
  const int b = blockIdx.x;
  const float *pp = partial + (size_t)b * n_blocks * 8;
  for (int k = 0; k < count; ++k) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_blocks; i += blockDim.x) acc += (double)pp[(size_t)i * 8 + k];
    const double tot = block_sum_d(acc, red);
    if (threadIdx.x == 0) out[(size_t)b * n_obs_total + obs_off + k] = (float)tot;
  }
}
// ---------------------------------------------------------------------------
// Meyer-Wallach (entanglement.py:69-103 with Tr rho_j^2 = a^2 + d^2 + 2 |c|^2): THREE reads of
// the state at n = 28 -- ceil((n - 12) / 8) + 1 in general -- instead of n.
//
// A read streams the state through 2^12-amplitude LDS tiles: the 4 lowest bits (128-byte rows)
// + two runs of 4 bit positions, [lo, lo+4) and [lo2, lo2+4), and reports the cross terms
// c_j = sum_{bit_j = 0} psi_i conj(psi_{i + 2^j}) of its bits.  The FIRST read's tile is 32 KiB
// of contiguous memory (bits 0..11); it also reports the signed populations of those 12 bits and
// the per-row totals from which the populations of EVERY other bit follow (sign of a row = one
// bit of its index), so the later reads carry cross terms only.
//
// Round-3 structure (tools/mw_tune.hip has the stand-alone bench it was tuned with):
//  * a lane's own 8 float4 span local bits {0, 9, 10, 11}: their cross terms (and, first read,
//    the 4-bit population butterfly) come straight out of the load registers, before staging;
//  * then ONE staging round trip and cross-only register gathers: local bits 1..4 and 5..8
//    (first read: 2 x 16 ds_read_b64) resp. 4..7 and, for bit 8 alone, 8 ds_read_b128 over
//    {0, 8, 9, 10} (later reads) -- 192 resp. 128 packed fmas per 16 amplitudes, nothing else;
//  * populations of the thread-mapped local bits 1..8 are the per-thread totals signed by the
//    thread index, applied once in the reduction at the end of a workgroup's walk;
//  * no register prefetch: 91 / 108 VGPRs = 5 / 4 workgroups per CU keep more bytes in flight
//    than a software pipeline at 3 (the round-2 kernel: 128 - 144 VGPRs, 4.1 - 4.85 TB/s);
//  * which positions share a tile matters: at n = 28 the pairs {12-15, 24-27} and {16-19, 20-23}
//    stream at 6.9 TB/s, {12-15, 20-23} at 5.7 and {20-27} at 4.5 (same bytes, same code;
//    profiles/r03_mw_tune.txt) -- mw_plan() pairs the 4-bit chunks outermost with innermost.
// Measured at n = 28 (MI355X): first read 0.34 ms, later reads 0.31 ms each; round 2: 0.52 +
// 2 x 0.44.
// ---------------------------------------------------------------------------
constexpr int kMwT = 12, kMwThreads = 256;
constexpr int kMwRowFirst = 48, kMwRowLater = 16;
constexpr int kMwRowLaterLow = 24;  // a later read that also reports positions 0..3: [16 + 2 p], [17 + 2 p]

struct MwReadArgs {
  const float2 *states;
  float *rows;  // [batch][rows_per_state][kMwRowFirst | kMwRowLater]
  int n, lo, lo2, q;  // tile = bits 0..3 + lo..lo+3 + lo2..lo2+3; 2^q tiles per workgroup
};

// (mw_cross: qmle_dev.h -- the tile kernels' fused Meyer-Wallach epilogue uses it too)
// cross terms of the 4 bits a 16-amplitude register gather spans
template <int B0>
__device__ __forceinline__ void mw_cross16(v2f (&cr)[12], const v2f (&r)[16]) {
  static_for<4>([&](auto t) {
    static_for<8>([&](auto pq) {
      constexpr int lowm = (1 << t) - 1;
      constexpr int c = (((int)pq & ~lowm) << 1) | ((int)pq & lowm);
      mw_cross(cr[B0 + (int)t], r[c], r[c | (1 << t)]);
    });
  });
}

// LOW (later reads behind a fused producing pass, round 5): the read also reports the cross terms of positions
// 0..3 -- bit 0 out of the halves of its own float4s, bits 1..3 out of one more 16-amplitude gather (local bits
// 1..4) -- which the producing pass then need not compute: that pass is bound by its vector arithmetic, a later
// read by HBM with ~55 % of its issue slots free (profiles/r05_mw_sq_resident.txt).
template <bool FIRST, bool NT, bool LOW = false>
__device__ __forceinline__ void mw_read_body(const MwReadArgs &a, float4 *smem4) {
  static_assert(!(FIRST && LOW), "the first read reports every low bit anyway");
  const uint32_t sbo = lds_offset_of(smem4);  // 0: no static LDS (launch side checks lds_base_is_zero)
  const uint32_t tid = threadIdx.x;
  const uint32_t jl = 2u * tid;  // local bits 1..8 from tid, 9..11 from u, bit 0 inside the float4
  const int lo = a.lo, lo2 = a.lo2;
  const uint64_t goff = ((uint64_t)(jl & 15u) | ((uint64_t)((jl >> 4) & 15u) << lo) | ((uint64_t)(jl >> 8) << lo2)) << 3;
  const uint64_t ustep = (uint64_t)1 << (lo2 + 1 + 3);  // local bit 9 = second run's bit 1
  const char *st = reinterpret_cast<const char *>(a.states + ((size_t)blockIdx.y << a.n)) + goff;
  const uint32_t slb = (sw(jl) << 3) + sbo;  // staging address of u = 0; u adds u << 12
  const uint32_t tile0 = blockIdx.x << a.q, n_it = 1u << a.q;
  const uint32_t r0 = lo - 4, r1 = lo2 - lo - 4;  // outer runs [4, lo), [lo+4, lo2), [lo2+4, n)

  // cross terms per reported bit: first read local bit b at cr[b]; later reads new bit k at cr[k]
  // (k < 4: lo + k, else lo2 + k - 4)
  v2f cr[12];
  static_for<12>([&](auto k) { cr[k] = (v2f){0.f, 0.f}; });
  float zin[4] = {0.f, 0.f, 0.f, 0.f}, tot = 0.f, zw[4] = {0.f, 0.f, 0.f, 0.f};
  // (LOW: positions 0..3 at cr[8 .. 11] -- the later reads use cr[0 .. 7] only)

  for (uint32_t it = 0; it < n_it; ++it) {
    const uint32_t t = tile0 + it;
    const uint64_t base = ((uint64_t)(t & ((1u << r0) - 1u)) << 4 | (uint64_t)((t >> r0) & ((1u << r1) - 1u)) << (lo + 4) |
                           (uint64_t)(t >> (r0 + r1)) << (lo2 + 4)) << 3;
    float4 v4[8];
    static_for<8>([&](auto u) { v4[u] = ld4<NT>(reinterpret_cast<const float4 *>(st + base + (uint64_t)u * ustep)); });
    v2f lo_[8], hi_[8];  // the two amplitudes of each float4
    static_for<8>([&](auto u) { lo_[u] = (v2f){v4[u].x, v4[u].y}; hi_[u] = (v2f){v4[u].z, v4[u].w}; });
    // ---- bits held by the lane's own 8 float4: local 0 (halves of a float4) and 9, 10, 11 (u) ----
    if (FIRST) {
      static_for<8>([&](auto u) { mw_cross(cr[0], lo_[u], hi_[u]); });
      float pr[16];
      static_for<8>([&](auto u) {
        const v2f q0 = lo_[u] * lo_[u], q1 = hi_[u] * hi_[u];
        pr[2 * u] = q0.x + q0.y;
        pr[2 * u + 1] = q1.x + q1.y;
      });
      float h0 = 0.f, h1 = 0.f, h2 = 0.f, s1[8], s2[4], s3[2];
      static_for<8>([&](auto i) { s1[i] = pr[2 * i] + pr[2 * i + 1]; h0 += pr[2 * i] - pr[2 * i + 1]; });
      static_for<4>([&](auto i) { s2[i] = s1[2 * i] + s1[2 * i + 1]; h1 += s1[2 * i] - s1[2 * i + 1]; });
      static_for<2>([&](auto i) { s3[i] = s2[2 * i] + s2[2 * i + 1]; h2 += s2[2 * i] - s2[2 * i + 1]; });
      zin[0] += h0; zin[1] += h1; zin[2] += h2; zin[3] += s3[0] - s3[1];
      const float tt = s3[0] + s3[1];
      tot += tt;
      static_for<4>([&](auto j) { zw[j] += __uint_as_float(__float_as_uint(tt) ^ (((it >> j) & 1u) << 31)); });
    }
    if (LOW) static_for<8>([&](auto u) { mw_cross(cr[8], lo_[u], hi_[u]); });
    static_for<3>([&](auto k) {
      constexpr int B = FIRST ? 9 + (int)k : 5 + (int)k;
      static_for<4>([&](auto pq) {
        constexpr int lowm = (1 << k) - 1;
        constexpr int u0 = (((int)pq & ~lowm) << 1) | ((int)pq & lowm);
        mw_cross(cr[B], lo_[u0], lo_[u0 | (1 << k)]);
        mw_cross(cr[B], hi_[u0], hi_[u0 | (1 << k)]);
      });
    });
    if (it) __syncthreads();  // the previous tile's gathers are done
    static_for<8>([&](auto u) { lds_st128(slb + ((uint32_t)u << 12), v4[u]); });
    __syncthreads();
    uint32_t tg = tid;
    asm volatile("" : "+v"(tg));  // keeps the gather addresses out of loop-carried registers
    {  // gather A: 16 amplitudes over local bits gA .. gA+3 (first read 1..4, later reads 4..7)
      constexpr int gA = FIRST ? 1 : 4;
      const uint32_t bs = (sw(ins0(ins0(ins0(ins0(tg, gA), gA + 1), gA + 2), gA + 3)) << 3) + sbo;
      v2f r[16];
      static_for<16>([&](auto c) {
        const u64 x = lds_ld64(bs ^ (sw((uint32_t)c << gA) << 3));
        r[c] = (v2f){__uint_as_float((uint32_t)x), __uint_as_float((uint32_t)(x >> 32))};
      });
      mw_cross16<FIRST ? gA : 0>(cr, r);
    }
    if (FIRST) {  // gather B: local bits 5..8
      constexpr int gB = 5;
      const uint32_t bs = (sw(ins0(ins0(ins0(ins0(tg, gB), gB + 1), gB + 2), gB + 3)) << 3) + sbo;
      v2f r[16];
      static_for<16>([&](auto c) {
        const u64 x = lds_ld64(bs ^ (sw((uint32_t)c << gB) << 3));
        r[c] = (v2f){__uint_as_float((uint32_t)x), __uint_as_float((uint32_t)(x >> 32))};
      });
      mw_cross16<gB>(cr, r);
    } else {
      if (LOW) {  // local bits 1..3 (positions 1..3): the gather over local bits 1..4, its first three bits
        constexpr int gL = 1;
        const uint32_t bs = (sw(ins0(ins0(ins0(ins0(tg, gL), gL + 1), gL + 2), gL + 3)) << 3) + sbo;
        v2f r[16];
        static_for<16>([&](auto c) {
          const u64 x = lds_ld64(bs ^ (sw((uint32_t)c << gL) << 3));
          r[c] = (v2f){__uint_as_float((uint32_t)x), __uint_as_float((uint32_t)(x >> 32))};
        });
        static_for<3>([&](auto t) {
          static_for<8>([&](auto pq) {
            constexpr int lowm = (1 << t) - 1;
            constexpr int c = (((int)pq & ~lowm) << 1) | ((int)pq & lowm);
            mw_cross(cr[9 + (int)t], r[c], r[c | (1 << t)]);
          });
        });
      }
      // local bit 8 alone: 8 float4 over local bits {0, 8, 9, 10}; thread index -> 1..7, 11
      const uint32_t e0 = ((tg & 127u) << 1) | ((tg >> 7) << 11);
      const uint32_t bs = (sw(e0) << 3) + sbo;
      float4 r[8];
      static_for<8>([&](auto c) {
        constexpr uint32_t e = (((uint32_t)c & 1u) << 8) | (((uint32_t)c >> 1) << 9);
        r[c] = lds_ld128(bs ^ (sw(e) << 3));
      });
      static_for<4>([&](auto pq) {
        mw_cross(cr[4], (v2f){r[2 * pq].x, r[2 * pq].y}, (v2f){r[2 * pq + 1].x, r[2 * pq + 1].y});
        mw_cross(cr[4], (v2f){r[2 * pq].z, r[2 * pq].w}, (v2f){r[2 * pq + 1].z, r[2 * pq + 1].w});
      });
    }
  }
  // ---- one reduction and one row per workgroup ----
  // first read: [0..23] cross terms of local bit b at 2b, 2b+1; [24..35] signed populations of
  // local bits 0..11; [36] total; [37..40] total signed by bit j of the tile's index in the walk.
  // later reads: [0..15] cross terms of new bit k at 2k, 2k+1.
  constexpr int NB = (FIRST || LOW) ? 12 : 8, NV = FIRST ? 41 : LOW ? kMwRowLaterLow : kMwRowLater;
  float red_v[NV];
  static_for<NB>([&](auto k) { red_v[2 * k] = cr[k].x; red_v[2 * k + 1] = cr[k].y; });
  if (FIRST) {
    red_v[24] = zin[0];
    static_for<8>([&](auto k) { red_v[25 + k] = ((tid >> k) & 1u) ? -tot : tot; });
    red_v[33] = zin[1]; red_v[34] = zin[2]; red_v[35] = zin[3];
    red_v[36] = tot;
    static_for<4>([&](auto j) { red_v[37 + j] = zw[j]; });
  }
  wave_sums_dpp63(red_v);
  __syncthreads();  // every gather has been read: the tile becomes scratch
  float *red = reinterpret_cast<float *>(smem4);
  const uint32_t lane = tid & (kWave - 1), w = tid / kWave;
  if (lane == kWave - 1) static_for<NV>([&](auto k) { red[w * NV + k] = red_v[k]; });
  __syncthreads();
  if (tid < NV) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMwThreads / kWave; ++i) s += red[i * NV + tid];
    a.rows[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (FIRST ? kMwRowFirst : LOW ? kMwRowLaterLow : kMwRowLater) + tid] = s;
  }
}
// Two entry points, because the occupancy that suits them differs: the later reads are bound by
// HBM alone and want every workgroup the LDS admits (5 per CU); the first read carries twice the
// arithmetic and runs FASTER at 4 workgroups per CU (0.33 vs 0.42 ms at n = 28, same
// instructions -- five workgroups of it fight over the vector pipe and the LDS between barriers).
template <bool NT>
__global__ void __launch_bounds__(kMwThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) k_mw_read_first(const MwReadArgs a) {
  extern __shared__ float4 smem4[];
  mw_read_body<true, NT>(a, smem4);
}
template <bool NT>
__global__ void __launch_bounds__(kMwThreads) k_mw_read_later(const MwReadArgs a) {
  extern __shared__ float4 smem4[];
  mw_read_body<false, NT>(a, smem4);
}
template <bool NT>
__global__ void __launch_bounds__(kMwThreads) k_mw_read_later_low(const MwReadArgs a) {
  extern __shared__ float4 smem4[];
  mw_read_body<false, NT, true>(a, smem4);
}

// Where the sums of bit position p come from: which later read (0 = the first read) and column.
struct MwPlan {
  int n_later;
  int lo[8], lo2[8], q[8];   // later reads
  int q_first;
  int src_read[QMLE_MAX_QUBITS], src_col[QMLE_MAX_QUBITS];
  uint32_t rows_first, rows_later[8];
};

// purity of one wire from the per-workgroup rows: one block per (state, bit position)
struct MwPurityArgs {
  const float *first;       // [batch][rows_first][kMwRowFirst]
  const float *later[8];    // [batch][rows_later[r]][kMwRowLater]
  uint32_t rows_first, rows_later[8];
  int q_first, n;
  int8_t src_read[QMLE_MAX_QUBITS], src_col[QMLE_MAX_QUBITS];
};
__global__ void __launch_bounds__(1024)
k_mw_purity(const MwPurityArgs a, float *__restrict__ pur_out /* [batch][n] by bit position */) {
  __shared__ double red[16];
  const int b = blockIdx.x, p = blockIdx.y;
  const float *fr = a.first + (size_t)b * a.rows_first * kMwRowFirst;
  double cr = 0, ci = 0, z = 0, tot = 0;
  for (uint32_t i = threadIdx.x; i < a.rows_first; i += blockDim.x) {
    const float *row = fr + (size_t)i * kMwRowFirst;
    const float t = row[36];
    tot += t;
    if (p < kMwT) { cr += row[2 * p]; ci += row[2 * p + 1]; z += row[24 + p]; }
    else if (p < kMwT + a.q_first) z += row[37 + p - kMwT];
    else z += ((i >> (p - kMwT - a.q_first)) & 1u) ? -(double)t : (double)t;
  }
  if (p >= kMwT) {
    const int r = a.src_read[p] - 1, col = a.src_col[p];
    const float *lr = a.later[r] + (size_t)b * a.rows_later[r] * kMwRowLater;
    for (uint32_t i = threadIdx.x; i < a.rows_later[r]; i += blockDim.x) {
      cr += lr[(size_t)i * kMwRowLater + 2 * col];
      ci += lr[(size_t)i * kMwRowLater + 2 * col + 1];
    }
  }
  cr = block_sum_d(cr, red);
  ci = block_sum_d(ci, red);
  z = block_sum_d(z, red);
  tot = block_sum_d(tot, red);
  if (threadIdx.x == 0) {
    const double pa = 0.5 * (tot + z), pd = 0.5 * (tot - z);
    pur_out[(size_t)b * a.n + p] = (float)(pa * pa + pd * pd + 2.0 * (cr * cr + ci * ci));
  }
}

__global__ void k_mw_tile_q(const float *__restrict__ pur, int n, int batch,
                            float *__restrict__ out, float *__restrict__ purities) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  double sum = 0.0;
  for (int p = 0; p < n; ++p) {
    const float v = pur[(size_t)b * n + p];
    sum += v;
    if (purities) purities[(size_t)b * n + (n - 1 - p)] = v;  // index by wire
  }
  out[b] = (float)(2.0 * (1.0 - sum / n));
}
// ---- Meyer-Wallach behind a producing pass (QMLE_MEAS_MEYER_WALLACH) -------------------------
// The circuit's last tile pass left one row of kMwFusedRowA floats per tile (tile_mw_row in
// qmle_tile_dev.h: cross terms and signed populations of the tile's T local bits + the tile's total);
// positions outside that tile take their cross terms from `later` reads of the stored state and their
// populations from the totals signed by the matching bit of the tile index.
constexpr int kMwFusedRowA = 48;  // = kMwFusedRow (qmle_tile_dev.h)
struct MwFusedArgs {
  const float *first;       // [batch][rows_first][kMwFusedRowA]
  const float *later[8];    // [batch][rows_later[r]][kMwRowLater]
  uint32_t rows_first, rows_later[8];
  uint32_t later_stride[8];  // floats per row of later read r (kMwRowLater, or kMwRowLaterLow for the read that reports 0..3)
  int lean;                  // positions 0..3: cross terms from later read 0 (columns 8 + p), not from the producing pass
  int T, n;
  int lg;                   // a row covers 2^lg consecutive tiles (the producing workgroup's walk): outer index
                            // bits < lg come with their own signed totals at [3T + 1 + i]
  int8_t loc[QMLE_MAX_QUBITS];        // position p -> local bit of the producing tile, or -1
  int8_t outer_idx[QMLE_MAX_QUBITS];  // position p -> bit of the tile index, or -1
  int8_t src_read[QMLE_MAX_QUBITS], src_col[QMLE_MAX_QUBITS];  // outer positions: later read (from 0) and its column
};
// gridDim.z = 1: the purity itself; > 1: slice z of the rows -> partial[b][p][z] = (cr, ci, z, tot) in
// fp64, summed by k_mw_purity_final (one block per (state, position) walking 32768 rows -- n = 28, one
// row per tile -- took 0.21 ms: only 28 workgroups were at it)
__global__ void __launch_bounds__(1024)
k_mw_purity_fused(const MwFusedArgs a, float *__restrict__ pur_out /* [batch][n] by bit position */,
                  double *__restrict__ partial) {
  __shared__ double red[16];
  const int b = blockIdx.x, p = blockIdx.y;
  const int T = a.T, j = a.loc[p], oi = a.outer_idx[p], lg = a.lg;
  const float *fr = a.first + (size_t)b * a.rows_first * kMwFusedRowA;
  const uint32_t per = (a.rows_first + gridDim.z - 1) / gridDim.z;
  const uint32_t r_lo = blockIdx.z * per, r_hi = r_lo + per < a.rows_first ? r_lo + per : a.rows_first;
  double cr = 0, ci = 0, z = 0, tot = 0;
  for (uint32_t i = r_lo + threadIdx.x; i < r_hi; i += blockDim.x) {
    const float *row = fr + (size_t)i * kMwFusedRowA;
    const float t = row[3 * T];
    tot += t;
    if (j >= 0) { cr += row[2 * j]; ci += row[2 * j + 1]; z += row[2 * T + j]; }  // (lean: zeros at 2 j for j < 4)
    else if (oi < lg) z += row[3 * T + 1 + oi];  // a bit of the tile's index inside the workgroup's walk
    else z += ((i >> (oi - lg)) & 1u) ? -(double)t : (double)t;
  }
  if (j < 0 || (a.lean && p < 4)) {
    const int r = a.src_read[p], col = a.src_col[p];
    const uint32_t stride = a.later_stride[r];
    const float *lr = a.later[r] + (size_t)b * a.rows_later[r] * stride;
    const uint32_t perl = (a.rows_later[r] + gridDim.z - 1) / gridDim.z;
    const uint32_t l_lo = blockIdx.z * perl, l_hi = l_lo + perl < a.rows_later[r] ? l_lo + perl : a.rows_later[r];
    for (uint32_t i = l_lo + threadIdx.x; i < l_hi; i += blockDim.x) {
      cr += lr[(size_t)i * stride + 2 * col];
      ci += lr[(size_t)i * stride + 2 * col + 1];
    }
  }
  cr = block_sum_d(cr, red);
  ci = block_sum_d(ci, red);
  z = block_sum_d(z, red);
  tot = block_sum_d(tot, red);
  if (threadIdx.x == 0) {
    if (gridDim.z > 1) {
      double *o = partial + (((size_t)b * a.n + p) * gridDim.z + blockIdx.z) * 4;
      o[0] = cr; o[1] = ci; o[2] = z; o[3] = tot;
    } else {
      const double pa = 0.5 * (tot + z), pd = 0.5 * (tot - z);
      pur_out[(size_t)b * a.n + p] = (float)(pa * pa + pd * pd + 2.0 * (cr * cr + ci * ci));
    }
  }
}
__global__ void __launch_bounds__(64)
k_mw_purity_final(const double *__restrict__ partial, int n, int batch, int slices, float *__restrict__ pur_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (state, position)
  if (i >= batch * n) return;
  double cr = 0, ci = 0, z = 0, tot = 0;
  for (int s = 0; s < slices; ++s) {
    const double *o = partial + ((size_t)i * slices + s) * 4;
    cr += o[0]; ci += o[1]; z += o[2]; tot += o[3];
  }
  const double pa = 0.5 * (tot + z), pd = 0.5 * (tot - z);
  pur_out[i] = (float)(pa * pa + pd * pd + 2.0 * (cr * cr + ci * ci));
}
// ---- the same sums in two coalesced steps (round 5; tiled producing passes) ---------------------
// k_mw_purity_fused walks the rows once per POSITION: 28 blocks re-read the 12.6 MB of rows of an n = 28 state
// through the L2 with 4 floats used of every 192-byte row (33 us + 11 us for the slices + 5 us to pack: 6 % of the
// fused call).  k_mw_colsum sums EVERY column of a row matrix in one coalesced sweep (thread = column, block = a
// slice of the rows; the totals signed by the bits of the row index ride along as extra "columns"), for the
// producing pass's rows and each later read's in one launch (blockIdx.z); k_mw_finish_cols turns a state's
// column sums into its purities and Q.
struct MwColsumArgs {
  const float *rows[9];   // [0] the producing pass's rows, [1 + r] later read r
  uint32_t n_rows[9], stride[9];
  int n_cols[9];
  int tot_col, n_signed;  // matrix 0 only: column of the row total, number of row-index bits to sign it by
  int slices;
  double *out[9];         // [batch][slices][n_cols + n_signed]
};
__global__ void __launch_bounds__(256)
k_mw_colsum(const MwColsumArgs a) {
  // thread = (column, one of four row lanes); eight rows in flight per thread -- with one load per iteration the
  // sweep was bound by the latency of 256 dependent round trips (0.13 ms at n = 28), with one row lane by 32
  __shared__ double part[4][64];
  const int m = blockIdx.z, b = blockIdx.y, sl = blockIdx.x, t = threadIdx.x;
  const int c = t & 63, lane = t >> 6;
  const int nc = a.n_cols[m], ns = m == 0 ? a.n_signed : 0;
  const uint32_t per = (a.n_rows[m] + a.slices - 1) / a.slices;
  const uint32_t lo = sl * per, hi = lo + per < a.n_rows[m] ? lo + per : a.n_rows[m];
  const float *base = a.rows[m] + (size_t)b * a.n_rows[m] * a.stride[m];
  double sum = 0.0;
  if (c < nc + ns) {
    const bool signedc = c >= nc;
    const int k = signedc ? c - nc : 0;
    const uint32_t cidx = signedc ? (uint32_t)a.tot_col : (uint32_t)c, stride = a.stride[m];
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t i = lo + (uint32_t)lane;
    for (; i + 28 < hi; i += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(i + 4 * u) * stride + cidx];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += (signedc && (((i + 4 * u) >> k) & 1u)) ? -(double)v[u] : (double)v[u];
    }
    for (; i < hi; i += 4) {
      const double v = (double)base[(size_t)i * stride + cidx];
      acc[0] += (signedc && ((i >> k) & 1u)) ? -v : v;
    }
    sum = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  }
  part[lane][c] = sum;
  __syncthreads();
  if (lane == 0 && c < nc + ns)
    a.out[m][((size_t)b * a.slices + sl) * (nc + ns) + c] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}
// one block per state: the slices of EVERY matrix are summed at once by all 1024 threads (thread = one of the <= 128
// columns of all matrices side by side x one of 8 slice lanes: coalesced, four loads in flight), then thread p = bit
// position turns the sums into its purity and a wave sum into Q.  (A first form with one thread per position walking
// the slices itself was latency-bound: 1800 dependent loads, 0.43 ms.)
constexpr int kMwColsMax = 128;
__global__ void __launch_bounds__(1024)
k_mw_finish_cols(const MwFusedArgs a, const MwColsumArgs c, int n_mats, float *__restrict__ out /* [batch][n + 1] */) {
  __shared__ double lane_sum[8][kMwColsMax];
  __shared__ double fin[kMwColsMax];
  __shared__ int off[10];
  const int b = blockIdx.x, t = threadIdx.x, col = t & (kMwColsMax - 1), q = t / kMwColsMax;
  if (t == 0) {
    int o = 0;
    for (int m = 0; m < n_mats; ++m) { off[m] = o; o += c.n_cols[m] + (m == 0 ? c.n_signed : 0); }
    off[n_mats] = o;
  }
  __syncthreads();
  {
    int m = 0;
    while (m + 1 < n_mats && col >= off[m + 1]) ++m;
    double acc = 0.0;
    if (col < off[n_mats]) {
      const int w = off[m + 1] - off[m];
      const double *o = c.out[m] + (size_t)b * c.slices * w + (col - off[m]);
      double a4[4] = {0, 0, 0, 0};
      int sl = q;
      for (; sl + 24 < c.slices; sl += 32) {  // four independent loads per round
#pragma unroll
        for (int u = 0; u < 4; ++u) a4[u] += o[(size_t)(sl + 8 * u) * w];
      }
      for (; sl < c.slices; sl += 8) a4[0] += o[(size_t)sl * w];
      acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    }
    lane_sum[q][col] = acc;
  }
  __syncthreads();
  if (q == 0) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += lane_sum[k][col];
    fin[col] = s;
  }
  __syncthreads();
  const int p = t, T = a.T, n = a.n, lg = a.lg;
  if (t >= kWave) return;
  double pur = 0.0;
  if (p < n) {
    const int j = a.loc[p], oi = a.outer_idx[p];
    const double tot = fin[3 * T];
    double cr = 0.0, ci = 0.0, z;
    if (j >= 0) {
      z = fin[2 * T + j];
      if (!(a.lean && p < 4)) { cr = fin[2 * j]; ci = fin[2 * j + 1]; }
    } else if (oi < lg) z = fin[3 * T + 1 + oi];
    else z = fin[c.n_cols[0] + (oi - lg)];
    if (j < 0 || (a.lean && p < 4)) {
      const int r = a.src_read[p], cc = a.src_col[p];
      cr += fin[off[1 + r] + 2 * cc];
      ci += fin[off[1 + r] + 2 * cc + 1];
    }
    const double pa = 0.5 * (tot + z), pd = 0.5 * (tot - z);
    pur = (double)(float)(pa * pa + pd * pd + 2.0 * (cr * cr + ci * ci));  // (rounded like k_mw_purity_fused's store)
    out[(size_t)b * (n + 1) + 1 + (n - 1 - p)] = (float)pur;  // index by wire
  }
  const double sum = wave_sum_d(pur);
  if (p == 0) out[(size_t)b * (n + 1)] = (float)(2.0 * (1.0 - sum / n));
}

// Whole-state plans: ONE row per state and every position local to the producing tile -- purities and Q of a
// state from its row in one work item (the same fp64 arithmetic as k_mw_purity_fused + k_mw_pack, whose
// 24 576 one-row workgroups took 16 + 5 us of the 12-qubit sampling loop's 185)
__global__ void __launch_bounds__(64)
k_mw_whole_state_finish(const MwFusedArgs a, int batch, float *__restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const float *row = a.first + (size_t)b * kMwFusedRowA;
  const int T = a.T, n = a.n;
  const double tot = (double)row[3 * T];
  float *o = out + (size_t)b * (n + 1);
  double sum = 0.0;
  for (int p = 0; p < n; ++p) {
    const int j = a.loc[p];
    const double cr = (double)row[2 * j], ci = (double)row[2 * j + 1], z = (double)row[2 * T + j];
    const double pa = 0.5 * (tot + z), pd = 0.5 * (tot - z);
    const float v = (float)(pa * pa + pd * pd + 2.0 * (cr * cr + ci * ci));
    sum += v;
    o[1 + (n - 1 - p)] = v;  // index by wire
  }
  o[0] = (float)(2.0 * (1.0 - sum / n));
}
// (Q [batch], purities by wire [batch][n]) -> out[b] = (Q, purities by wire)
__global__ void k_mw_pack_wires(const float *__restrict__ q, const float *__restrict__ pur, int n, int batch,
                                float *__restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  float *o = out + (size_t)b * (n + 1);
  o[0] = q[b];
  for (int w = 0; w < n; ++w) o[1 + w] = pur[(size_t)b * n + w];
}
// out[b] = (Q, purity of wire 0, ..., purity of wire n-1)
__global__ void k_mw_pack(const float *__restrict__ pur, int n, int batch, float *__restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  double sum = 0.0;
  float *o = out + (size_t)b * (n + 1);
  for (int p = 0; p < n; ++p) {
    const float v = pur[(size_t)b * n + p];
    sum += v;
    o[1 + (n - 1 - p)] = v;  // index by wire
  }
  o[0] = (float)(2.0 * (1.0 - sum / n));
}

// vec(rho) measurements: rho[i][j] at flat index i * D + j (ket bits first)
__global__ void __launch_bounds__(256)
k_density_probs(const float2 *__restrict__ rho, int n, float *__restrict__ out) {
  const int b = blockIdx.y;
  const uint64_t D = (uint64_t)1 << n;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D) out[(size_t)b * D + i] = rho[((size_t)b << (2 * n)) + i * (D + 1)].x;
}

__global__ void __launch_bounds__(256)
k_density_expval(const float2 *__restrict__ rho, int n, ObsBits obs, int n_obs,
                 float *__restrict__ out) {
  __shared__ double red[16];
  const int b = blockIdx.x, k = blockIdx.y;
  const uint64_t D = (uint64_t)1 << n;
  const float2 *r = rho + ((size_t)b << (2 * n));
  const int p = obs.bits[k];
  double acc = 0.0;
  for (uint64_t i = threadIdx.x; i < D; i += blockDim.x) {
    const float v = r[i * (D + 1)].x;
    acc += ((i >> p) & 1ull) ? -(double)v : (double)v;
  }
  const double tot = block_sum_d(acc, red);
  if (threadIdx.x == 0) out[(size_t)b * n_obs + k] = (float)tot;
}

}  // namespace

namespace qmle {

int expval_blocks(int n) {
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  const uint64_t seg = (uint64_t)kEzThreads * kEzUnroll;
  uint64_t n_seg = (chunks + seg - 1) / seg;
  if (n_seg > 2048) n_seg = 2048;
  return (int)n_seg;
}

int run_expval(const float2 *states, int n, int batch, const int8_t *obs_bits, int n_obs,
               float *d_out, void *ws, size_t ws_bytes, hipStream_t stream) {
  if (n_obs < 1 || n_obs > QMLE_MAX_QUBITS) return QMLE_ERR_INVALID_ARG;
  const int nb = expval_blocks(n);
  const size_t need = (size_t)batch * nb * (QMLE_MAX_QUBITS + 1) * sizeof(float);
  if (ws_bytes < need) return QMLE_ERR_WORKSPACE;
  ObsBits ob;
  for (int k = 0; k < QMLE_MAX_QUBITS; ++k) ob.row_mask[k] = 0u;
  for (int k = 0; k < n_obs; ++k) {
    if (obs_bits[k] < 0 || obs_bits[k] >= n) return QMLE_ERR_WIRE_RANGE;
    ob.bits[k] = obs_bits[k];
  }
  hipLaunchKernelGGL(k_expval_partial, dim3(nb, batch), dim3(kEzThreads), 0, stream,
                     reinterpret_cast<const float4 *>(states), n, (float *)ws);
  hipLaunchKernelGGL(k_expval_final, dim3(batch, n_obs), dim3(256), 0, stream,
                     (const float *)ws, nb, n_obs, ob, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int overlap_blocks(int n) {
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  uint64_t b = (chunks + 256 * 4 - 1) / (256 * 4);
  if (b < 1) b = 1;
  if (b > 1024) b = 1024;
  return (int)b;
}

// Z-parity observables (bit-position masks) of resident states: the stand-alone kernels
int run_parity_pos(const float2 *states, int n, int batch, const uint32_t *pos_masks,
                          int n_obs, float *d_out, void *ws, size_t ws_bytes,
                          hipStream_t stream) {
  const int nb = overlap_blocks(n);
  if (ws_bytes < (size_t)batch * nb * 8 * sizeof(float)) return QMLE_ERR_WORKSPACE;
  for (int o0 = 0; o0 < n_obs; o0 += 8) {
    ParityMasks pm;
    pm.count = n_obs - o0 < 8 ? n_obs - o0 : 8;
    for (int k = 0; k < 8; ++k) pm.m[k] = k < pm.count ? pos_masks[o0 + k] : 0u;
    hipLaunchKernelGGL(k_parity_partial, dim3(nb, batch), dim3(256), 0, stream,
                       (const float4 *)states, n, pm, (float *)ws);
    hipLaunchKernelGGL(k_parity_final, dim3(batch), dim3(nb >= 256 ? 256 : 64), 0, stream,
                       (const float *)ws, nb, pm.count, n_obs, o0, d_out);
  }
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

void launch_expval_final(const float *partial, int n_rows, int batch, int n_obs, const ObsBits &ob,
                         float *d_out, hipStream_t stream) {
  hipLaunchKernelGGL(k_expval_final, dim3(batch, n_obs), dim3(256), 0, stream, partial, n_rows, n_obs, ob, d_out);
}

void launch_probs(const float2 *states, float *d_out, uint64_t total_chunks, hipStream_t stream) {
  hipLaunchKernelGGL(k_probs, dim3(grid_for(total_chunks, 256)), dim3(256), 0, stream,
                     reinterpret_cast<const float4 *>(states), reinterpret_cast<float2 *>(d_out), total_chunks);
}

void launch_density(const float2 *states, float2 *d_out, int n, int batch, hipStream_t stream) {
  const uint64_t D = (uint64_t)1 << n;
  hipLaunchKernelGGL(k_density, dim3(grid_for(D * D, 256, 1u << 20), batch), dim3(256), 0, stream, states, d_out, n);
}

}  // namespace qmle

extern "C" {

size_t qmle_expval_workspace_bytes(int n_qubits, int batch) {
  if (n_qubits < 1 || batch < 1) return 0;
  return (size_t)batch * expval_blocks(n_qubits) * (QMLE_MAX_QUBITS + 1) * sizeof(float) + 256;
}

int qmle_expval_z(const void *d_states, int n_qubits, int batch, const int32_t *obs_wires,
                  int n_obs, float *d_out, void *d_workspace, size_t workspace_bytes,
                  qmle_stream stream) {
  if (!d_states || !d_out || !d_workspace || !obs_wires || n_qubits < 1 ||
      n_qubits > QMLE_MAX_QUBITS || batch < 1 || batch > 65535)
    return QMLE_ERR_INVALID_ARG;
  if (n_obs < 1 || n_obs > QMLE_MAX_QUBITS) return QMLE_ERR_INVALID_ARG;
  int8_t bits[QMLE_MAX_QUBITS];
  for (int k = 0; k < n_obs; ++k) {
    if (obs_wires[k] < 0 || obs_wires[k] >= n_qubits) return QMLE_ERR_WIRE_RANGE;
    bits[k] = (int8_t)(n_qubits - 1 - obs_wires[k]);
  }
  return run_expval((const float2 *)d_states, n_qubits, batch, bits, n_obs, d_out,
                    d_workspace, workspace_bytes, (hipStream_t)stream);
}

int qmle_probs(const void *d_states, int n_qubits, int batch, float *d_out, qmle_stream stream) {
  if (!d_states || !d_out || n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS || batch < 1)
    return QMLE_ERR_INVALID_ARG;
  const uint64_t tc = (uint64_t)batch << (n_qubits - 1);
  hipLaunchKernelGGL(k_probs, dim3(grid_for(tc, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float4 *)d_states, (float2 *)d_out, tc);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_density(const void *d_states, int n_qubits, int batch, void *d_out, qmle_stream stream) {
  if (!d_states || !d_out || n_qubits < 1 || batch < 1 || batch > 65535) return QMLE_ERR_INVALID_ARG;
  if (n_qubits > 15) return QMLE_ERR_UNSUPPORTED;
  const uint64_t D = (uint64_t)1 << n_qubits;
  hipLaunchKernelGGL(k_density, dim3(grid_for(D * D, 256, 1u << 20), batch), dim3(256), 0,
                     (hipStream_t)stream, (const float2 *)d_states, (float2 *)d_out, n_qubits);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_marginal_probs(const void *d_states, int n_qubits, int batch, const int32_t *keep_wires,
                        int n_keep, float *d_out, qmle_stream stream_) {
  if (!d_states || !d_out || !keep_wires || n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS ||
      batch < 1 || batch > 65535 || n_keep < 1 || n_keep > n_qubits || n_keep > 24)
    return QMLE_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  // kept wires stay in ascending wire order regardless of `keep` order
  // (jaqsi.py:141-146): output bit k (LSB first) <- the k-th LARGEST wire.
  uint64_t mask = 0;
  for (int k = 0; k < n_keep; ++k) {
    if (keep_wires[k] < 0 || keep_wires[k] >= n_qubits) return QMLE_ERR_WIRE_RANGE;
    if (mask & (1ull << keep_wires[k])) return QMLE_ERR_DUPLICATE_WIRES;
    mask |= 1ull << keep_wires[k];
  }
  KeepBits kb;
  kb.n_keep = n_keep;
  int k = 0;
  for (int w = n_qubits - 1; w >= 0; --w)
    if (mask & (1ull << w)) kb.bits[k++] = (int8_t)(n_qubits - 1 - w);
  HIPCHK(hipMemsetAsync(d_out, 0, ((size_t)batch << n_keep) * sizeof(float), stream));
  const uint64_t D = (uint64_t)1 << n_qubits;
  if (n_keep <= 12) {
    // (<= 512 workgroups per state: >= 2^n / 512 terms are summed in LDS per global atomic)
    hipLaunchKernelGGL(k_marginal_lds, dim3(grid_for(D, 256, 512), batch), dim3(256),
                       sizeof(float) << n_keep, stream, (const float2 *)d_states, d_out, n_qubits, kb);
  } else {
    hipLaunchKernelGGL(k_marginal, dim3(grid_for(D, 256, 4096), batch), dim3(256), 0, stream,
                       (const float2 *)d_states, d_out, n_qubits, kb);
  }
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

size_t qmle_pair_fidelity_workspace_bytes(int n_qubits, int n_pairs) {
  if (n_qubits < 1 || n_pairs < 1) return 0;
  return (size_t)n_pairs * overlap_blocks(n_qubits) * sizeof(float2) + 256;
}

int qmle_pair_fidelity(const void *d_states, int n_qubits, int n_pairs, float *d_out,
                       void *d_workspace, size_t workspace_bytes, qmle_stream stream_) {
  if (!d_states || !d_out || !d_workspace || n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS ||
      n_pairs < 1)
    return QMLE_ERR_INVALID_ARG;
  const int nb = overlap_blocks(n_qubits);
  if (workspace_bytes < (size_t)n_pairs * nb * sizeof(float2)) return QMLE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const uint64_t chunks = (uint64_t)1 << (n_qubits - 1);
  if (n_qubits <= 16 && n_pairs >= 64) {  // enough pairs to fill the chip with one workgroup each
    const int threads = chunks >= 256 ? 256 : 64;
    hipLaunchKernelGGL(k_pair_fidelity_small, dim3(n_pairs), dim3(threads), 0, stream,
                       (const float4 *)d_states, n_qubits, n_pairs, d_out);
    HIPCHK(hipGetLastError());
    return QMLE_OK;
  }
  for (int p0 = 0; p0 < n_pairs; p0 += 65535) {
    const int pc = n_pairs - p0 < 65535 ? n_pairs - p0 : 65535;
    // pairs (i, i + n_pairs): shift both halves by p0
    hipLaunchKernelGGL(k_overlap_partial, dim3(nb, pc), dim3(256), 0, stream,
                       (const float4 *)d_states + (size_t)p0 * chunks, n_qubits, n_pairs,
                       (float2 *)d_workspace + (size_t)p0 * nb);
  }
  hipLaunchKernelGGL(k_overlap_final, dim3(n_pairs), dim3(nb >= 256 ? 256 : 64), 0, stream,
                     (const float2 *)d_workspace, nb, n_pairs, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_density_probs(const void *d_rho, int n_qubits, int batch, float *d_out,
                       qmle_stream stream) {
  if (!d_rho || !d_out || n_qubits < 1 || 2 * n_qubits > QMLE_MAX_QUBITS || batch < 1 ||
      batch > 65535)
    return QMLE_ERR_INVALID_ARG;
  const uint64_t D = (uint64_t)1 << n_qubits;
  hipLaunchKernelGGL(k_density_probs, dim3(grid_for(D, 256), batch), dim3(256), 0,
                     (hipStream_t)stream, (const float2 *)d_rho, n_qubits, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_density_expval_z(const void *d_rho, int n_qubits, int batch, const int32_t *obs_wires,
                          int n_obs, float *d_out, qmle_stream stream) {
  if (!d_rho || !d_out || !obs_wires || n_qubits < 1 || 2 * n_qubits > QMLE_MAX_QUBITS ||
      batch < 1 || batch > 65535 || n_obs < 1 || n_obs > QMLE_MAX_QUBITS)
    return QMLE_ERR_INVALID_ARG;
  ObsBits ob;
  for (int k = 0; k < n_obs; ++k) {
    if (obs_wires[k] < 0 || obs_wires[k] >= n_qubits) return QMLE_ERR_WIRE_RANGE;
    ob.bits[k] = (int8_t)(n_qubits - 1 - obs_wires[k]);
  }
  hipLaunchKernelGGL(k_density_expval, dim3(batch, n_obs), dim3(256), 0, (hipStream_t)stream,
                     (const float2 *)d_rho, n_qubits, ob, n_obs, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

size_t qmle_overlap_workspace_bytes(int n_qubits, int count) {
  if (n_qubits < 1 || count < 1) return 0;
  return (size_t)count * overlap_blocks(n_qubits) * sizeof(float2) + 256;
}

int qmle_overlap(const void *d_a, const void *d_b, int n_qubits, int count, void *d_out,
                 void *d_workspace, size_t workspace_bytes, qmle_stream stream_) {
  if (!d_a || !d_b || !d_out || !d_workspace || n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS ||
      count < 1)
    return QMLE_ERR_INVALID_ARG;
  const int nb = overlap_blocks(n_qubits);
  if (workspace_bytes < (size_t)count * nb * sizeof(float2)) return QMLE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const uint64_t chunks = (uint64_t)1 << (n_qubits - 1);
  for (int p0 = 0; p0 < count; p0 += 65535) {
    const int pc = count - p0 < 65535 ? count - p0 : 65535;
    hipLaunchKernelGGL(k_overlap2_partial, dim3(nb, pc), dim3(256), 0, stream,
                       (const float4 *)d_a + (size_t)p0 * chunks,
                       (const float4 *)d_b + (size_t)p0 * chunks, n_qubits,
                       (float2 *)d_workspace + (size_t)p0 * nb);
  }
  hipLaunchKernelGGL(k_overlap2_final, dim3(count), dim3(nb >= 256 ? 256 : 64), 0, stream,
                     (const float2 *)d_workspace, nb, count, (float2 *)d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

size_t qmle_expval_parity_workspace_bytes(int n_qubits, int batch) {
  if (n_qubits < 1 || batch < 1) return 0;
  return (size_t)batch * overlap_blocks(n_qubits) * 8 * sizeof(float) + 256;
}

int qmle_expval_parity(const void *d_states, int n_qubits, int batch, const uint32_t *wire_masks,
                       int n_obs, float *d_out, void *d_workspace, size_t workspace_bytes,
                       qmle_stream stream_) {
  if (!d_states || !d_out || !d_workspace || !wire_masks || n_qubits < 1 ||
      n_qubits > QMLE_MAX_QUBITS || batch < 1 || batch > 65535 || n_obs < 1)
    return QMLE_ERR_INVALID_ARG;
  const int nb = overlap_blocks(n_qubits);
  if (workspace_bytes < (size_t)batch * nb * 8 * sizeof(float)) return QMLE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  for (int o0 = 0; o0 < n_obs; o0 += 8) {
    ParityMasks pm;
    pm.count = n_obs - o0 < 8 ? n_obs - o0 : 8;
    for (int k = 0; k < 8; ++k) {
      uint32_t bits = 0;
      if (k < pm.count) {
        const uint32_t wm = wire_masks[o0 + k];  // bit w set <=> wire w in the parity
        if (n_qubits < 32 && (wm >> n_qubits)) return QMLE_ERR_WIRE_RANGE;
        for (int w = 0; w < n_qubits; ++w)
          if (wm & (1u << w)) bits |= 1u << (n_qubits - 1 - w);
      }
      pm.m[k] = bits;
    }
    hipLaunchKernelGGL(k_parity_partial, dim3(nb, batch), dim3(256), 0, stream,
                       (const float4 *)d_states, n_qubits, pm, (float *)d_workspace);
    hipLaunchKernelGGL(k_parity_final, dim3(batch), dim3(nb >= 256 ? 256 : 64), 0, stream,
                       (const float *)d_workspace, nb, pm.count, n_obs, o0, d_out);
  }
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

}  // extern "C"

// Plan of the reads for an n-qubit state (host only).  Positions >= 12 are cut into 4-bit chunks
// [12,16), [16,20), ... with the last one aligned to the top [n-4, n); a later read takes two
// chunks, paired outermost with innermost -- at n = 28: {12-15, 24-27} and {16-19, 20-23}, the
// pairs that stream fastest (see the note above k_mw_read).  An odd chunk is paired with a
// lower, already reported one.  A chunk may overlap its neighbour (n not a multiple of 4): every
// position takes its sums from the first read that reports it.
static MwPlan mw_plan(int n, int batch) {
  MwPlan pl;
  std::memset(&pl, 0, sizeof(pl));
  for (int p = 0; p < n; ++p) pl.src_read[p] = -1;
  for (int p = 0; p < kMwT && p < n; ++p) { pl.src_read[p] = 0; pl.src_col[p] = p; }
  const uint32_t tiles = 1u << (n - kMwT);
  // tiles per workgroup: the first read keeps ~40 sums per work item, a long walk amortises its
  // reduction (2^4); the later reads stream best at 2^2 (tools/mw_tune.hip); never fewer than
  // ~2048 workgroups per launch
  auto pick_q = [&](int want) {
    int q = 0;
    while (q < want && (((uint64_t)batch * tiles) >> (q + 1)) >= 2048) ++q;
    return q;
  };
  static const int q_env = std::getenv("QMLE_MW_Q") ? atoi(std::getenv("QMLE_MW_Q")) : -1;
  pl.q_first = pick_q(4);
  if (q_env >= 0 && q_env <= 4 && (tiles >> q_env) >= 1) pl.q_first = q_env;
  pl.rows_first = tiles >> pl.q_first;
  int chunks[8], nc = 0;
  for (int c = kMwT; c < n; c += 4) chunks[nc++] = c + 4 <= n ? c : n - 4;
  int i = 0, j = nc - 1;
  while (i <= j) {
    int a = chunks[i], b = i < j ? chunks[j] : -1;
    if (b < 0) {  // odd one out: pair it with a lower, already reported chunk
      b = a;
      a = b >= 16 ? 8 : b - 4;
    }
    if (a > b) std::swap(a, b);
    if (b < a + 4) a = b - 4;  // overlapping chunks (n not a multiple of 4): shift the lower one down
    const int r = pl.n_later++;
    pl.lo[r] = a;
    pl.lo2[r] = b;
    pl.q[r] = pick_q(2);
    if (q_env >= 0 && q_env <= 4 && (tiles >> q_env) >= 1) pl.q[r] = q_env < 2 ? q_env : 2;
    pl.rows_later[r] = tiles >> pl.q[r];
    for (int k = 0; k < 8; ++k) {
      const int p = k < 4 ? a + k : b + k - 4;
      if (p < n && pl.src_read[p] < 0) { pl.src_read[p] = r + 1; pl.src_col[p] = k; }
    }
    ++i;
    --j;
  }
  return pl;
}

namespace qmle {

// Later reads that cover the positions outside the producing tile (`tile_mask`: bit p set <=> position p
// is a local bit of that tile; it holds positions 0..3).  A later read reports the cross terms of two
// runs of four positions [lo, lo + 4), [lo2, lo2 + 4) with 4 <= lo, lo + 4 <= lo2, lo2 + 4 <= n.
struct MwCover {
  int n_later = 0;
  int lo[8], lo2[8], q[8];
  uint32_t rows_later[8];
  int src_read[QMLE_MAX_QUBITS], src_col[QMLE_MAX_QUBITS];
  bool ok = true;
};
constexpr int kMwCoverPairing = 1;  // measured at n = 28 behind the K2-style last tile {0..6, 23..27}: 0.792 ms after the circuit (0: 0.801, 2: 0.801; profiles/r05_mw_pairing_nt.txt)
static MwCover mw_cover(int n, uint32_t tile_mask, int batch) {
  MwCover cv;
  for (int p = 0; p < n; ++p) cv.src_read[p] = cv.src_col[p] = -1;
  if ((tile_mask & 0xFu) != 0xFu || n < kMwT) { cv.ok = false; return cv; }
  int chunks[8], nc = 0, covered_to = 0;
  for (int p = 4; p < n; ++p) {
    if ((tile_mask >> p) & 1u) continue;
    if (p < covered_to) continue;
    if (nc == 8) { cv.ok = false; return cv; }
    const int c = p + 4 <= n ? p : n - 4;
    chunks[nc++] = c;
    covered_to = c + 4;
  }
  const uint32_t tiles = 1u << (n - kMwT);
  auto pick_q = [&](int want) {
    int q = 0;
    while (q < want && (((uint64_t)batch * tiles) >> (q + 1)) >= 2048) ++q;
    return q;
  };
  // (four runs: which two share a read decides how the read streams -- tools/mw_lean_ab.py with QMLE_MW_PAIRING=0/1/2:
  // 0 = outermost with innermost (the stand-alone reads' rule), 1 = neighbours, 2 = alternate)
  if (nc == 4) {
    const char *e = std::getenv("QMLE_MW_PAIRING");
    const int mode = e ? atoi(e) : kMwCoverPairing;
    if (mode == 1) std::swap(chunks[1], chunks[3]);        // (0,1) (2,3): loop pairs (c0,c3') = (0,1), (c1',c2) = (3,2)
    else if (mode == 2) std::swap(chunks[2], chunks[3]);   // (0,2) (1,3)
  }
  int i = 0, j = nc - 1;
  while (i <= j) {
    int a = chunks[i], b = i < j ? chunks[j] : -1;
    if (b < 0) {  // odd one out: any other run completes the read
      b = a;
      a = b >= 16 ? 8 : b >= 8 ? b - 4 : -1;
      if (a < 0) { a = b; b = a + 4; if (b + 4 > n) { cv.ok = false; return cv; } }
    }
    if (a > b) std::swap(a, b);
    if (b < a + 4) a = b - 4;  // overlapping runs (the last one is clamped to n - 4): shift the lower one down
    if (a < 4) { cv.ok = false; return cv; }
    const int r = cv.n_later++;
    cv.lo[r] = a;
    cv.lo2[r] = b;
    cv.q[r] = pick_q(2);
    cv.rows_later[r] = tiles >> cv.q[r];
    for (int k = 0; k < 8; ++k) {
      const int p = k < 4 ? a + k : b + k - 4;
      if (p < n && !((tile_mask >> p) & 1u) && cv.src_read[p] < 0) { cv.src_read[p] = r; cv.src_col[p] = k; }
    }
    ++i;
    --j;
  }
  for (int p = 0; p < n; ++p)
    if (!((tile_mask >> p) & 1u) && cv.src_read[p] < 0) cv.ok = false;
  return cv;
}

static int mw_colsum_slices(uint32_t most_rows) {  // >= 64 rows per block, <= 256 blocks per state and matrix
  int slices = 1;
  while (slices < 256 && (most_rows >> 6) > (uint32_t)slices) slices *= 2;
  return slices;
}

static uint32_t stage_tile_mask(const Stage &st) {
  uint32_t m = 0;
  for (int j = 0; j < st.T; ++j) m |= 1u << st.tile_bits[j];
  return m;
}

// Can stage `last` (the plan's last one) report its tile's Meyer-Wallach sums, and do fewer reads
// than the stand-alone kernel's remain?  (16 amplitudes per work item: T >= 10.)
bool mw_fusable(int n, const Stage &last) {
  if (last.kind != ST_TILE || last.T < 10 || last.T > 14) return false;
  if (last.T == n) return true;
  const MwCover cv = mw_cover(n, stage_tile_mask(last), 1);
  return cv.ok && cv.n_later < qmle_meyer_wallach_reads(n);
}

// Tiled state: positions 0..3 sit in the producing tile AND in the tile of every later read; the later reads are
// bound by HBM, the producing pass by its arithmetic -- so the first later read reports them (QMLE_MW_NO_LEAN=1: A/B).
bool mw_lean(int n, const Stage &last) {
  if (last.kind != ST_TILE || last.T >= n || last.T < 10) return false;
  if (std::getenv("QMLE_MW_NO_LEAN") != nullptr) return false;  // (read per call)
  for (int j = 0; j < 4; ++j)
    if (last.tile_bits[j] != j) return false;
  const MwCover cv = mw_cover(n, stage_tile_mask(last), 1);
  return cv.ok && cv.n_later >= 1;
}

size_t mw_fused_ws_bytes(int n, int batch, const Stage &last) {
  size_t fl = ((size_t)1 << (n - last.T)) * kMwFusedRowA;  // one row per tile
  if (last.T < n) {
    const MwCover cv = mw_cover(n, stage_tile_mask(last), batch);
    for (int r = 0; r < cv.n_later; ++r) fl += (size_t)cv.rows_later[r] * kMwRowLaterLow;  // (either row length fits)
  }
  size_t colsum = 0;  // k_mw_colsum's slices (tiled producing passes only)
  if (last.T < n) {
    const MwCover cv = mw_cover(n, stage_tile_mask(last), batch);
    uint32_t most = 1u << (n - last.T);
    for (int r = 0; r < cv.n_later; ++r) most = std::max(most, cv.rows_later[r]);
    colsum = (size_t)batch * mw_colsum_slices(most) * (kMwFusedRowA + QMLE_MAX_QUBITS + (size_t)cv.n_later * kMwRowLaterLow) *
                 sizeof(double) + 64;
  }
  return ((size_t)batch * fl + (size_t)batch * QMLE_MAX_QUBITS) * sizeof(float) +
         (size_t)8192 * 4 * sizeof(double) + 1024 + colsum;  // + the purity kernel's slices (< 8192 (state, position, slice) sums)
}

// rows_first = ws (filled by the producing pass: [batch][tiles][kMwFusedRowA]); the later reads' rows and
// the purities follow it.  d_out [batch][n + 1] = (Q, purities by wire).
int run_mw_fused(const float2 *states, int n, int batch, const Stage &last, int row_shift, void *ws_,
                 size_t ws_bytes, float *d_out, hipStream_t stream) {
  if (batch < 1 || batch > 65535) return QMLE_ERR_INVALID_ARG;
  if (ws_bytes < mw_fused_ws_bytes(n, batch, last) - 512) return QMLE_ERR_WORKSPACE;
  float *ws = (float *)ws_;
  MwFusedArgs pa;
  std::memset(&pa, 0, sizeof(pa));
  pa.first = ws;
  pa.rows_first = (1u << (n - last.T)) >> row_shift;
  pa.lg = row_shift;
  pa.T = last.T;
  pa.n = n;
  for (int p = 0; p < QMLE_MAX_QUBITS; ++p) pa.loc[p] = pa.outer_idx[p] = pa.src_read[p] = pa.src_col[p] = -1;
  for (int j = 0; j < last.T; ++j) pa.loc[(int)last.tile_bits[j]] = (int8_t)j;
  for (int i = 0; i < n - last.T; ++i) pa.outer_idx[(int)last.outer_bits[i]] = (int8_t)i;
  ws += (size_t)batch * ((size_t)1 << (n - last.T)) * kMwFusedRowA;  // (one row per tile)
  if (last.T < n) {
    const MwCover cv = mw_cover(n, stage_tile_mask(last), batch);
    if (!cv.ok) return QMLE_ERR_INTERNAL;
    if (FirstUse once{6}; once.first) {
      QMLE_LDS_BASE_CHECK(k_mw_read_later<true>);
      QMLE_LDS_BASE_CHECK(k_mw_read_later<false>);
      QMLE_LDS_BASE_CHECK(k_mw_read_later_low<true>);
      QMLE_LDS_BASE_CHECK(k_mw_read_later_low<false>);
      once.done();
    }
    const uint32_t tiles = 1u << (n - kMwT);
    const bool nt = ((uint64_t)batch << (n + 3)) >= (1ull << 30);
    pa.lean = mw_lean(n, last) ? 1 : 0;
    for (int r = 0; r < cv.n_later; ++r) {
      MwReadArgs a;
      a.states = states;
      a.n = n;
      a.rows = ws;
      a.lo = cv.lo[r];
      a.lo2 = cv.lo2[r];
      a.q = cv.q[r];
      const bool low = pa.lean && r == 0;  // the first later read also reports positions 0..3
      pa.later[r] = ws;
      pa.rows_later[r] = cv.rows_later[r];
      pa.later_stride[r] = low ? kMwRowLaterLow : kMwRowLater;
      ws += (size_t)batch * cv.rows_later[r] * pa.later_stride[r];
      const dim3 grid(tiles >> a.q, batch);
      if (low) {
        if (nt) hipLaunchKernelGGL(k_mw_read_later_low<true>, grid, dim3(kMwThreads), (size_t)8 << kMwT, stream, a);
        else hipLaunchKernelGGL(k_mw_read_later_low<false>, grid, dim3(kMwThreads), (size_t)8 << kMwT, stream, a);
      } else if (nt) hipLaunchKernelGGL(k_mw_read_later<true>, grid, dim3(kMwThreads), (size_t)8 << kMwT, stream, a);
      else hipLaunchKernelGGL(k_mw_read_later<false>, grid, dim3(kMwThreads), (size_t)8 << kMwT, stream, a);
    }
    for (int p = 0; p < n; ++p) { pa.src_read[p] = (int8_t)cv.src_read[p]; pa.src_col[p] = (int8_t)cv.src_col[p]; }
    if (pa.lean)
      for (int p = 0; p < 4; ++p) { pa.src_read[p] = 0; pa.src_col[p] = (int8_t)(8 + p); }
  }
  if (last.T == n && pa.rows_first == 1) {
    hipLaunchKernelGGL(k_mw_whole_state_finish, dim3((batch + 63) / 64), dim3(64), 0, stream, pa, batch, d_out);
    HIPCHK(hipGetLastError());
    return QMLE_OK;
  }
  static const bool old_purity = std::getenv("QMLE_MW_OLD_PURITY") != nullptr;  // (A/B: the per-position walk)
  int colsum_w0 = kMwFusedRowA + (n - last.T - pa.lg > 0 ? n - last.T - pa.lg : 0), colsum_w = colsum_w0;
  for (int r = 0; r < 8 && pa.later[r]; ++r) colsum_w += (int)pa.later_stride[r];
  if (last.T < n && !old_purity && 3 * last.T + 5 <= kMwFusedRowA && colsum_w0 <= 64 && colsum_w <= kMwColsMax) {
    MwColsumArgs ca;
    std::memset(&ca, 0, sizeof(ca));
    uint32_t most = pa.rows_first;
    for (int r = 0; r < 8; ++r) most = std::max(most, pa.rows_later[r]);
    const int slices = mw_colsum_slices(most);
    ca.slices = slices;
    ca.rows[0] = pa.first; ca.n_rows[0] = pa.rows_first; ca.stride[0] = kMwFusedRowA; ca.n_cols[0] = kMwFusedRowA;
    ca.tot_col = 3 * last.T;
    ca.n_signed = n - last.T - pa.lg > 0 ? n - last.T - pa.lg : 0;
    int n_mats = 1;
    for (int r = 0; r < 8 && pa.later[r]; ++r) {
      ca.rows[1 + r] = pa.later[r]; ca.n_rows[1 + r] = pa.rows_later[r]; ca.stride[1 + r] = pa.later_stride[r];
      ca.n_cols[1 + r] = (int)pa.later_stride[r];
      n_mats = 2 + r;
    }
    double *dp = (double *)(((uintptr_t)ws + 7) & ~(uintptr_t)7);
    for (int m = 0; m < n_mats; ++m) {
      ca.out[m] = dp;
      dp += (size_t)batch * slices * (ca.n_cols[m] + (m == 0 ? ca.n_signed : 0));
    }
    if ((size_t)((char *)dp - (char *)ws_) > ws_bytes) return QMLE_ERR_WORKSPACE;
    hipLaunchKernelGGL(k_mw_colsum, dim3(slices, batch, n_mats), dim3(256), 0, stream, ca);
    hipLaunchKernelGGL(k_mw_finish_cols, dim3(batch), dim3(1024), 0, stream, pa, ca, n_mats, d_out);
    HIPCHK(hipGetLastError());
    return QMLE_OK;
  }
  float *d_pur = ws;
  ws += (size_t)batch * QMLE_MAX_QUBITS;
  // enough workgroups for the row sums: slices of >= 256 rows, <= 64 per (state, position)
  uint32_t most = pa.rows_first;
  for (int r = 0; r < 8; ++r) most = std::max(most, pa.rows_later[r]);
  int slices = 1;
  while (slices < 64 && (most >> 8) > (uint32_t)slices && (uint64_t)batch * n * slices < 4096) slices *= 2;
  const int threads = most / slices >= 1024 ? 1024 : most / slices >= 256 ? 256 : 64;
  double *d_part = (double *)(((uintptr_t)ws + 7) & ~(uintptr_t)7);
  hipLaunchKernelGGL(k_mw_purity_fused, dim3(batch, n, slices), dim3(threads), 0, stream, pa, d_pur, d_part);
  if (slices > 1)
    hipLaunchKernelGGL(k_mw_purity_final, dim3((batch * n + 63) / 64), dim3(64), 0, stream, (const double *)d_part, n,
                       batch, slices, d_pur);
  hipLaunchKernelGGL(k_mw_pack, dim3((batch + 63) / 64), dim3(64), 0, stream, (const float *)d_pur, n, batch, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

// resident states, no producing pass to lean on: qmle_meyer_wallach, packed like run_mw_fused
int run_mw_resident(const float2 *states, int n, int batch, void *ws, size_t ws_bytes, float *d_out,
                    hipStream_t stream) {
  const size_t need = qmle_meyer_wallach_workspace_bytes(n, batch);
  if (ws_bytes < need + (size_t)batch * (n + 1) * sizeof(float)) return QMLE_ERR_WORKSPACE;
  float *q = (float *)((char *)ws + ((need + 255) & ~(size_t)255));
  float *pur = q + batch;
  const int rc = qmle_meyer_wallach(states, n, batch, q, pur, ws, need, (qmle_stream)stream);
  if (rc != QMLE_OK) return rc;
  hipLaunchKernelGGL(k_mw_pack_wires, dim3((batch + 63) / 64), dim3(64), 0, stream, (const float *)q,
                     (const float *)pur, n, batch, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

size_t mw_resident_ws_bytes(int n, int batch) {
  return ((qmle_meyer_wallach_workspace_bytes(n, batch) + 255) & ~(size_t)255) + (size_t)batch * (n + 2) * sizeof(float) + 256;
}

}  // namespace qmle

extern "C" {

int qmle_meyer_wallach_reads(int n_qubits) {
  if (n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS) return 0;
  // below the tile size: one (cache-resident) sweep per wire
  return n_qubits >= kMwT ? 1 + mw_plan(n_qubits, 1).n_later : n_qubits;
}

size_t qmle_meyer_wallach_workspace_bytes(int n_qubits, int batch) {
  if (n_qubits < 1 || batch < 1) return 0;
  if (n_qubits >= kMwT) {
    const MwPlan pl = mw_plan(n_qubits, batch);
    size_t fl = (size_t)pl.rows_first * kMwRowFirst;
    for (int r = 0; r < pl.n_later; ++r) fl += (size_t)pl.rows_later[r] * kMwRowLater;
    return ((size_t)batch * fl + (size_t)batch * QMLE_MAX_QUBITS) * sizeof(float) + 512;
  }
  return (size_t)batch * n_qubits * overlap_blocks(n_qubits) * sizeof(float4) + 256;
}

int qmle_meyer_wallach(const void *d_states, int n_qubits, int batch, float *d_out,
                       float *d_purities, void *d_workspace, size_t workspace_bytes,
                       qmle_stream stream_) {
  if (!d_states || !d_out || !d_workspace || n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS ||
      batch < 1 || batch > 65535)
    return QMLE_ERR_INVALID_ARG;
  if (workspace_bytes + 256 < qmle_meyer_wallach_workspace_bytes(n_qubits, batch))
    return QMLE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int n = n_qubits;
  if (n >= kMwT) {  // LDS-staged tiles: 1 + ceil((n - 12) / 8) reads of the state
    if (FirstUse once{5}; once.first) {
      QMLE_LDS_BASE_CHECK(k_mw_read_first<true>);
      QMLE_LDS_BASE_CHECK(k_mw_read_first<false>);
      QMLE_LDS_BASE_CHECK(k_mw_read_later<true>);
      QMLE_LDS_BASE_CHECK(k_mw_read_later<false>);
      once.done();
    }
    const MwPlan pl = mw_plan(n, batch);
    for (int p = 0; p < n; ++p)
      if (pl.src_read[p] < 0) return QMLE_ERR_INTERNAL;
    const uint32_t tiles = 1u << (n - kMwT);
    // >= 1 GiB per launch: stream past the caches
    const bool nt = ((uint64_t)batch << (n + 3)) >= (1ull << 30);
    const size_t lds = (size_t)8 << kMwT;
    MwPurityArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    float *ws = (float *)d_workspace;
    MwReadArgs a;
    a.states = (const float2 *)d_states;
    a.n = n;
    a.rows = ws;
    a.lo = 4;
    a.lo2 = 8;
    a.q = pl.q_first;
    pa.first = ws;
    pa.rows_first = pl.rows_first;
    pa.q_first = pl.q_first;
    pa.n = n;
    ws += (size_t)batch * pl.rows_first * kMwRowFirst;
    // The reads are independent of each other.  The first read is the arithmetic-heavy one (192
    // packed fmas + the population butterfly per 16 amplitudes) and is the one that suffers when
    // it starts on a chip that has just idled through the tiny reduction kernels of a previous
    // call (0.33 ms warm, 0.42 - 0.47 ms cold at n = 28); the later reads are bound by HBM
    // alone.  So it runs LAST (QMLE_MW_FIRST_FIRST=1 for the A/B).
    static const bool first_first = std::getenv("QMLE_MW_FIRST_FIRST") != nullptr;
    auto launch_first = [&]() {
      const dim3 grid(tiles >> a.q, batch);
      if (nt) hipLaunchKernelGGL(k_mw_read_first<true>, grid, dim3(kMwThreads), lds, stream, a);
      else hipLaunchKernelGGL(k_mw_read_first<false>, grid, dim3(kMwThreads), lds, stream, a);
    };
    const MwReadArgs a_first = a;
    if (first_first || pl.n_later == 0) launch_first();
    for (int r = 0; r < pl.n_later; ++r) {
      a.rows = ws;
      a.lo = pl.lo[r];
      a.lo2 = pl.lo2[r];
      a.q = pl.q[r];
      pa.later[r] = ws;
      pa.rows_later[r] = pl.rows_later[r];
      ws += (size_t)batch * pl.rows_later[r] * kMwRowLater;
      const dim3 grid(tiles >> a.q, batch);
      if (nt) hipLaunchKernelGGL(k_mw_read_later<true>, grid, dim3(kMwThreads), lds, stream, a);
      else hipLaunchKernelGGL(k_mw_read_later<false>, grid, dim3(kMwThreads), lds, stream, a);
    }
    if (!first_first && pl.n_later > 0) {
      a = a_first;
      launch_first();
    }
    for (int p = 0; p < n; ++p) { pa.src_read[p] = (int8_t)pl.src_read[p]; pa.src_col[p] = (int8_t)pl.src_col[p]; }
    float *d_pur = ws;
    hipLaunchKernelGGL(k_mw_purity, dim3(batch, n), dim3(pl.rows_first >= 1024 ? 1024 : pl.rows_first >= 256 ? 256 : 64),
                       0, stream, pa, d_pur);
    hipLaunchKernelGGL(k_mw_tile_q, dim3((batch + 63) / 64), dim3(64), 0, stream,
                       (const float *)d_pur, n, batch, d_out, d_purities);
    HIPCHK(hipGetLastError());
    return QMLE_OK;
  }
  const int nb = overlap_blocks(n_qubits);
  for (int p = 0; p < n_qubits; ++p)
    hipLaunchKernelGGL(k_cross_partial, dim3(nb, batch), dim3(256), 0, stream,
                       (const float4 *)d_states, n_qubits, p, (float4 *)d_workspace, nb);
  hipLaunchKernelGGL(k_mw_final, dim3(batch), dim3(nb >= 256 ? 256 : 64), 0, stream,
                     (const float4 *)d_workspace, n_qubits, nb, batch, d_out, d_purities);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_philox_uniform_f32_device(const uint64_t key[2], uint64_t n, double low, double high, float *d_out,
                                   qmle_stream stream_) {
  if (!key || (!d_out && n > 0) || n > (1ull << 40)) return QMLE_ERR_INVALID_ARG;
  if (n == 0) return QMLE_OK;
  hipLaunchKernelGGL(k_philox_uniform, dim3(grid_for((n + 3) / 4, 256, 1u << 16)), dim3(256), 0, (hipStream_t)stream_,
                     key[0], key[1], n, low, high - low, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_philox_uniform_f32_device_key(const uint64_t *d_key, uint64_t n, double low, double high, float *d_out,
                                       qmle_stream stream_) {
  if (!d_key || (!d_out && n > 0) || n > (1ull << 40)) return QMLE_ERR_INVALID_ARG;
  if (n == 0) return QMLE_OK;
  hipLaunchKernelGGL(k_philox_uniform_devkey, dim3(grid_for((n + 3) / 4, 256, 1u << 16)), dim3(256), 0,
                     (hipStream_t)stream_, d_key, n, low, high - low, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_histogram(const float *d_values, int64_t count, int n_bins, float lo, float hi,
                   int32_t *d_counts, qmle_stream stream_) {
  if (!d_values || !d_counts || count < 0 || n_bins < 1 || !(hi > lo)) return QMLE_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (n_bins <= 4096 && count > 0 && count <= (1 << 16)) {  // one workgroup, one launch
    const int threads = count >= 1024 ? 1024 : count >= 256 ? 256 : 64;
    hipLaunchKernelGGL(k_histogram_lds<true>, dim3(1), dim3(threads), (size_t)n_bins * sizeof(int), stream,
                       d_values, count, n_bins, lo, hi, d_counts);
    HIPCHK(hipGetLastError());
    return QMLE_OK;
  }
  HIPCHK(hipMemsetAsync(d_counts, 0, (size_t)n_bins * sizeof(int32_t), stream));
  if (count > 0) {
    if (n_bins <= 4096)
      hipLaunchKernelGGL(k_histogram_lds<false>, dim3(grid_for((uint64_t)count, 1024 * 16, 1024)), dim3(1024),
                         (size_t)n_bins * sizeof(int), stream, d_values, count, n_bins, lo, hi, d_counts);
    else
      hipLaunchKernelGGL(k_histogram, dim3(grid_for((uint64_t)count, 256, 1024)), dim3(256), 0,
                         stream, d_values, count, n_bins, lo, hi, d_counts);
  }
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

size_t qmle_sample_workspace_bytes(int n_qubits, int batch) {
  if (n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS || batch < 1) return 0;
  return ((size_t)batch << n_qubits) * sizeof(double);
}

int qmle_sample_counts(const float *d_probs, int n_qubits, int batch, int shots, uint64_t seed,
                       uint64_t row_offset, int32_t *d_counts, float *d_est_probs,
                       void *d_workspace, size_t workspace_bytes, qmle_stream stream_) {
  if (!d_probs || !d_counts || !d_workspace || n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS ||
      batch < 1 || batch > 65535 || shots < 1)
    return QMLE_ERR_INVALID_ARG;
  if (workspace_bytes < qmle_sample_workspace_bytes(n_qubits, batch))
    return QMLE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const uint64_t D = (uint64_t)1 << n_qubits;
  double *cdf = (double *)d_workspace;
  HIPCHK(hipMemsetAsync(d_counts, 0, (size_t)batch * D * sizeof(int32_t), stream));
  hipLaunchKernelGGL(k_cdf, dim3(batch), dim3(256), 0, stream, d_probs, D, cdf);
  const unsigned gx = (unsigned)(((int64_t)shots + kShotsPerBlock - 1) / kShotsPerBlock);
  if (D <= (uint64_t)kLdsHistMax)
    hipLaunchKernelGGL(k_sample<true>, dim3(gx, batch), dim3(256), 0, stream, cdf, D, shots,
                       seed, row_offset, d_counts);
  else
    hipLaunchKernelGGL(k_sample<false>, dim3(gx, batch), dim3(256), 0, stream, cdf, D, shots,
                       seed, row_offset, d_counts);
  if (d_est_probs)
    hipLaunchKernelGGL(k_counts_to_probs, dim3(grid_for((uint64_t)batch * D, 256, 4096)),
                       dim3(256), 0, stream, d_counts, (uint64_t)batch * D, 1.0f / (float)shots,
                       d_est_probs);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

int qmle_probs_diag_expval(const float *d_probs, int n_qubits, int batch,
                           const int32_t *obs_wires, const int32_t *obs_n_wires,
                           const int32_t *obs_diag_off, const float *d_diag, int n_obs,
                           float *d_out, void *d_workspace, size_t workspace_bytes,
                           qmle_stream stream_) {
  if (!d_probs || !d_out || !obs_wires || !obs_n_wires || !obs_diag_off || !d_workspace ||
      n_qubits < 1 || n_qubits > QMLE_MAX_QUBITS || batch < 1 || batch > 65535 || n_obs < 1 ||
      n_obs > 65535)
    return QMLE_ERR_INVALID_ARG;
  if (workspace_bytes < (size_t)n_obs * sizeof(DiagObs)) return QMLE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  std::vector<DiagObs> host(n_obs);
  int w0 = 0;
  for (int k = 0; k < n_obs; ++k) {
    const int nw = obs_n_wires[k];
    if (nw < 1 || nw > n_qubits) return QMLE_ERR_INVALID_ARG;
    uint32_t seen = 0;
    for (int j = 0; j < nw; ++j) {
      const int w = obs_wires[w0 + j];
      if (w < 0 || w >= n_qubits) return QMLE_ERR_WIRE_RANGE;
      if (seen & (1u << w)) return QMLE_ERR_DUPLICATE_WIRES;
      seen |= 1u << w;
      host[k].bits[j] = (int8_t)(n_qubits - 1 - w);
    }
    host[k].n_wires = nw;
    host[k].diag_off = obs_diag_off[k];
    if (host[k].diag_off >= 0 && !d_diag) return QMLE_ERR_INVALID_ARG;
    w0 += nw;
  }
  HIPCHK(hipMemcpyAsync(d_workspace, host.data(), (size_t)n_obs * sizeof(DiagObs),
                        hipMemcpyHostToDevice, stream));
  HIPCHK(hipStreamSynchronize(stream));  // `host` dies with this frame
  hipLaunchKernelGGL(k_probs_diag_expval, dim3(batch, n_obs), dim3(256), 0, stream, d_probs,
                     (uint64_t)1 << n_qubits, (const DiagObs *)d_workspace, d_diag, n_obs, d_out);
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

size_t qmle_probs_diag_expval_workspace_bytes(int n_obs) {
  return n_obs < 1 ? 0 : (size_t)n_obs * sizeof(DiagObs);
}

}  // extern "C"
