// libqmle_sv, streaming passes: one (controlled) 2x2 gate / the Golomb diagonal / fills, in place in HBM.
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
// direct (HBM-streaming) controlled-2x2 kernel: one float4 = 2 amplitudes per
// access, partners found by zero-bit insertion; in place.
//   MODE 0: no control, target bit >= 1      MODE 1: no control, target bit 0
//   MODE 2: control >= 1, target >= 1        MODE 3: control >= 1, target bit 0
//   MODE 4: control bit 0, target >= 1
// ---------------------------------------------------------------------------

// One work item per thread and an exact grid: a persistent grid-stride loop measured
// 15-20 % slower for this in-place two-stream pattern (tools/k1_tune.hip).
template <int MODE, bool DIAG, bool NT>
__global__ void __launch_bounds__(256)
k_direct_1q(float4 *__restrict__ states, int n, int pt, int pc,
            const float *__restrict__ mats, uint32_t mat_floats, uint32_t mat_off,
            uint64_t items, uint32_t blk_mul) {
  const int b = blockIdx.y;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  float4 *st = states + (size_t)b * chunks;
  // blk_mul (odd, power-of-two grids only; 0 = off): workgroup i takes block (i * blk_mul) mod grid -- the
  // workgroups in flight at any time are then spread over the whole state instead of one window of it
  const uint32_t blk = blk_mul ? (blockIdx.x * blk_mul) & (gridDim.x - 1u) : blockIdx.x;
  const uint64_t k = (uint64_t)blk * 256u + threadIdx.x;
  if ((MODE < 5 || MODE == 9) && k >= items) return;  // modes 5 .. 8: exact grids, whole waves
  const Mat2 m = load_mat2(mats + (size_t)b * mat_floats + mat_off);
  if constexpr (MODE == 0) {
    if constexpr (DIAG) {  // items = all chunks
      float4 v = ld4<NT>(st + k);
      const float2 f = ((k >> (pt - 1)) & 1ull) ? m.m11 : m.m00;
      const float2 x = cmul(f, make_float2(v.x, v.y)), y = cmul(f, make_float2(v.z, v.w));
      st4<NT>(st + k, make_float4(x.x, x.y, y.x, y.y));
    } else {
      const uint64_t c0 = ins0_64(k, pt - 1), c1 = c0 | (1ull << (pt - 1));
      float4 v0 = ld4<NT>(st + c0), v1 = ld4<NT>(st + c1);
      float2 a0 = make_float2(v0.x, v0.y), a1 = make_float2(v1.x, v1.y);
      float2 b0 = make_float2(v0.z, v0.w), b1 = make_float2(v1.z, v1.w);
      apply2(m, a0, a1);
      apply2(m, b0, b1);
      st4<NT>(st + c0, make_float4(a0.x, a0.y, b0.x, b0.y));
      st4<NT>(st + c1, make_float4(a1.x, a1.y, b1.x, b1.y));
    }
  } else if constexpr (MODE == 1) {
    float4 v = ld4<NT>(st + k);
    float2 a0 = make_float2(v.x, v.y), a1 = make_float2(v.z, v.w);
    if constexpr (DIAG) {
      a0 = cmul(m.m00, a0);
      a1 = cmul(m.m11, a1);
    } else {
      apply2(m, a0, a1);
    }
    st4<NT>(st + k, make_float4(a0.x, a0.y, a1.x, a1.y));
  } else if constexpr (MODE == 2) {
    if constexpr (DIAG) {  // items = chunks with control bit set
      const uint64_t c = ins0_64(k, pc - 1) | (1ull << (pc - 1));
      float4 v = ld4<NT>(st + c);
      const float2 f = ((c >> (pt - 1)) & 1ull) ? m.m11 : m.m00;
      const float2 x = cmul(f, make_float2(v.x, v.y)), y = cmul(f, make_float2(v.z, v.w));
      st4<NT>(st + c, make_float4(x.x, x.y, y.x, y.y));
    } else {
      const int lo = pt < pc ? pt - 1 : pc - 1, hi = pt < pc ? pc - 1 : pt - 1;
      const uint64_t c0 = ins0_64(ins0_64(k, lo), hi) | (1ull << (pc - 1));
      const uint64_t c1 = c0 | (1ull << (pt - 1));
      float4 v0 = ld4<NT>(st + c0), v1 = ld4<NT>(st + c1);
      float2 a0 = make_float2(v0.x, v0.y), a1 = make_float2(v1.x, v1.y);
      float2 b0 = make_float2(v0.z, v0.w), b1 = make_float2(v1.z, v1.w);
      apply2(m, a0, a1);
      apply2(m, b0, b1);
      st4<NT>(st + c0, make_float4(a0.x, a0.y, b0.x, b0.y));
      st4<NT>(st + c1, make_float4(a1.x, a1.y, b1.x, b1.y));
    }
  } else if constexpr (MODE == 3) {
    const uint64_t c = ins0_64(k, pc - 1) | (1ull << (pc - 1));
    float4 v = ld4<NT>(st + c);
    float2 a0 = make_float2(v.x, v.y), a1 = make_float2(v.z, v.w);
    if constexpr (DIAG) {
      a0 = cmul(m.m00, a0);
      a1 = cmul(m.m11, a1);
    } else {
      apply2(m, a0, a1);
    }
    st4<NT>(st + c, make_float4(a0.x, a0.y, a1.x, a1.y));
  } else if constexpr (MODE == 5) {
    // uncontrolled dense gate on bit 1..6: the partner chunk sits in lane ^ 2^(pt-1) of the same
    // wave.  Every lane loads and stores contiguous float4s (coalesced like the diagonal gate)
    // and fetches the partner's through the cross-lane path; it computes its own half of the
    // pair only.  items = chunks / 2, two rows per lane.  (tools/k1_tune.hip: 0.76 -> 0.70 ms)
    const uint64_t c0 = (uint64_t)blk * 512u + threadIdx.x, c1 = c0 + 256u;
    const bool up = (threadIdx.x >> (pt - 1)) & 1u;
    const float2 ms = up ? m.m11 : m.m00, mo = up ? m.m10 : m.m01;
    float4 v[2] = {ld4<NT>(st + c0), ld4<NT>(st + c1)};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float4 o;
      o.x = __shfl_xor(v[u].x, 1 << (pt - 1), kWave); o.y = __shfl_xor(v[u].y, 1 << (pt - 1), kWave);
      o.z = __shfl_xor(v[u].z, 1 << (pt - 1), kWave); o.w = __shfl_xor(v[u].w, 1 << (pt - 1), kWave);
      const float2 x = cfma(mo, make_float2(o.x, o.y), cmul(ms, make_float2(v[u].x, v[u].y)));
      const float2 y = cfma(mo, make_float2(o.z, o.w), cmul(ms, make_float2(v[u].z, v[u].w)));
      v[u] = make_float4(x.x, x.y, y.x, y.y);
    }
    st4<NT>(st + c0, v[0]);
    st4<NT>(st + c1, v[1]);
  } else if constexpr (MODE == 7) {
    // controlled gate with the control on bits 0..3 and the target on bits 1..6: both live inside
    // the 1 KiB a wave covers with one float4 per lane, and every 128-byte line holds both
    // control values, so all 16 D bytes move whatever the kernel does.  Stream them like the
    // diagonal gate -- one contiguous float4 per lane in, one out --, fetch the partner through
    // the cross-lane path and rewrite only the amplitudes whose control bit is set.  (A single-
    // gate LDS tile pass did this at 0.757 ms for n = 28; 8 D accounting: 0.35 -> 0.39.)
    const uint64_t c = (uint64_t)blk * 256u + threadIdx.x;  // items = all chunks, exact grid
    const bool up = (threadIdx.x >> (pt - 1)) & 1u;
    const float2 ms = up ? m.m11 : m.m00, mo = up ? m.m10 : m.m01;
    const float4 v = ld4<NT>(st + c);
    float4 o;
    o.x = __shfl_xor(v.x, 1 << (pt - 1), kWave); o.y = __shfl_xor(v.y, 1 << (pt - 1), kWave);
    o.z = __shfl_xor(v.z, 1 << (pt - 1), kWave); o.w = __shfl_xor(v.w, 1 << (pt - 1), kWave);
    const bool lane_ctl = pc == 0 ? true : ((threadIdx.x >> (pc - 1)) & 1u) != 0;
    float2 x = make_float2(v.x, v.y), y = make_float2(v.z, v.w);
    if (lane_ctl) {
      if (pc != 0) x = cfma(mo, make_float2(o.x, o.y), cmul(ms, x));  // control bit 0: only the odd amplitude
      y = cfma(mo, make_float2(o.z, o.w), cmul(ms, y));
    }
    st4<NT>(st + c, make_float4(x.x, x.y, y.x, y.y));
  } else if constexpr (MODE == 6) {
    // uncontrolled dense gate on a high bit (>= 21): a wave takes 4 ADJACENT rows of each of the
    // two streams (4 KiB contiguous per stream), all loads of one stream first: the DRAM banks
    // see fewer alternations between the two rows 2^pt amplitudes apart.  items = pairs / 4.
    // (tools/k1_tune.hip: 0.75-0.79 -> 0.70 ms for bits 21..27)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t row0 = ((uint64_t)blk * 4u + wave) * 4u;
    float4 v0[4], v1[4];
    uint64_t c0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { c0[u] = ins0_64((row0 + u) * 64u + lane, pt - 1); v0[u] = ld4<NT>(st + c0[u]); }
#pragma unroll
    for (int u = 0; u < 4; ++u) v1[u] = ld4<NT>(st + (c0[u] | (1ull << (pt - 1))));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float2 a0 = make_float2(v0[u].x, v0[u].y), a1 = make_float2(v1[u].x, v1[u].y);
      float2 b0 = make_float2(v0[u].z, v0[u].w), b1 = make_float2(v1[u].z, v1[u].w);
      apply2(m, a0, a1);
      apply2(m, b0, b1);
      v0[u] = make_float4(a0.x, a0.y, b0.x, b0.y);
      v1[u] = make_float4(a1.x, a1.y, b1.x, b1.y);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) st4<NT>(st + c0[u], v0[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) st4<NT>(st + (c0[u] | (1ull << (pt - 1))), v1[u]);
  } else if constexpr (MODE == 8) {
    // controlled dense gate, control and target both on chunk bits >= 8: mode 6's bursts inside the
    // control = 1 half -- a wave takes 4 adjacent rows (4 KiB contiguous) of each of the two streams, all
    // loads of one stream first.  items = controlled pairs / 4.
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t row0 = ((uint64_t)blk * 4u + wave) * 4u;
    const int lo = pt < pc ? pt - 1 : pc - 1, hi = pt < pc ? pc - 1 : pt - 1;
    float4 v0[4], v1[4];
    uint64_t c0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c0[u] = ins0_64(ins0_64((row0 + u) * 64u + lane, lo), hi) | (1ull << (pc - 1));
      v0[u] = ld4<NT>(st + c0[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) v1[u] = ld4<NT>(st + (c0[u] | (1ull << (pt - 1))));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float2 a0 = make_float2(v0[u].x, v0[u].y), a1 = make_float2(v1[u].x, v1[u].y);
      float2 b0 = make_float2(v0[u].z, v0[u].w), b1 = make_float2(v1[u].z, v1[u].w);
      apply2(m, a0, a1);
      apply2(m, b0, b1);
      v0[u] = make_float4(a0.x, a0.y, b0.x, b0.y);
      v1[u] = make_float4(a1.x, a1.y, b1.x, b1.y);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) st4<NT>(st + c0[u], v0[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) st4<NT>(st + (c0[u] | (1ull << (pt - 1))), v1[u]);
  } else if constexpr (MODE == 9) {
    // controlled PHASE (CZ, ControlledPhaseShift: diag(1, 1, 1, e^{i phi}), operations.py:1100, 1171-1201), control
    // and target both on chunk bits: only the |11> quarter of the state changes -- 4 D bytes read, 4 D written
    // (SURVEY 8-d), where the generic controlled-diagonal path (mode 2) rewrites the whole control = 1 half.
    // items = chunks / 4.
    const int lo = pt < pc ? pt - 1 : pc - 1, hi = pt < pc ? pc - 1 : pt - 1;
    const uint64_t c = ins0_64(ins0_64(k, lo), hi) | (1ull << (pc - 1)) | (1ull << (pt - 1));
    const float4 v = ld4<NT>(st + c);
    const float2 x = cmul(m.m11, make_float2(v.x, v.y)), y = cmul(m.m11, make_float2(v.z, v.w));
    st4<NT>(st + c, make_float4(x.x, x.y, y.x, y.y));
  } else {  // MODE 4: control is the in-chunk bit -> only the odd amplitude
    if constexpr (DIAG) {  // items = all chunks
      float4 v = ld4<NT>(st + k);
      const float2 f = ((k >> (pt - 1)) & 1ull) ? m.m11 : m.m00;
      const float2 y = cmul(f, make_float2(v.z, v.w));
      st4<NT>(st + k, make_float4(v.x, v.y, y.x, y.y));
    } else {
      const uint64_t c0 = ins0_64(k, pt - 1), c1 = c0 | (1ull << (pt - 1));
      float4 v0 = ld4<NT>(st + c0), v1 = ld4<NT>(st + c1);
      float2 b0 = make_float2(v0.z, v0.w), b1 = make_float2(v1.z, v1.w);
      apply2(m, b0, b1);
      st4<NT>(st + c0, make_float4(v0.x, v0.y, b0.x, b0.y));
      st4<NT>(st + c1, make_float4(v1.x, v1.y, b1.x, b1.y));
    }
  }
}

__global__ void __launch_bounds__(256)
k_diag_all(float4 *__restrict__ states, int n, const float *__restrict__ marks,
           const float *__restrict__ angles, int n_slots, int slot) {
  const int b = blockIdx.y;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  float4 *st = states + (size_t)b * chunks;
  const float x = angles[(size_t)b * n_slots + slot];
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < chunks; k += stride) {
    float4 v = st[k];
    float s0, c0, s1, c1;
    sincosf(marks[2 * k] * x, &s0, &c0);
    sincosf(marks[2 * k + 1] * x, &s1, &c1);
    const float2 a = cmul(make_float2(c0, -s0), make_float2(v.x, v.y));
    const float2 c = cmul(make_float2(c1, -s1), make_float2(v.z, v.w));
    st[k] = make_float4(a.x, a.y, c.x, c.y);
  }
}

__global__ void __launch_bounds__(256)
k_init_zero(float4 *__restrict__ states, int n) {
  const int b = blockIdx.y;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  float4 *st = states + (size_t)b * chunks;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < chunks; k += stride)
    st[k] = make_float4(k == 0 ? 1.f : 0.f, 0.f, 0.f, 0.f);
}

// Zero fill of `count` float4 (the all-live initialising pass: every tile but tile 0 of a state is
// zeros).  ONE plain store per thread and no loop: 6.79 TB/s on 4 GiB; four stores per thread
// 6.29, sixteen 5.71, a grid-stride loop 5.35, hipMemsetAsync 6.59, non-temporal stores a little
// below each (tools/fill_bench.hip) -- and one workgroup per 32 KiB tile inside k_tile2 6.0.
template <bool NT>
__global__ void __launch_bounds__(256) k_fill_zero(float4 *__restrict__ p, uint64_t count) {
  const uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (k < count) st4<NT>(p + k, make_float4(0.f, 0.f, 0.f, 0.f));
}

constexpr bool kFillNtDefault = false;

template <int MODE>
void launch_direct_mode(bool diag, bool nt, dim3 grid, hipStream_t stream, float4 *st, int n,
                        int pt, int pc, const float *mats, uint32_t mat_floats,
                        uint32_t mat_off, uint64_t items, uint32_t blk_mul) {
#define QMLE_LAUNCH_DIRECT(D, N)                                                              \
  hipLaunchKernelGGL((k_direct_1q<MODE, D, N>), grid, dim3(256), 0, stream, st, n, pt, pc, mats, \
                     mat_floats, mat_off, items, blk_mul)
  if (diag) { if (nt) QMLE_LAUNCH_DIRECT(true, true); else QMLE_LAUNCH_DIRECT(true, false); }
  else { if (nt) QMLE_LAUNCH_DIRECT(false, true); else QMLE_LAUNCH_DIRECT(false, false); }
#undef QMLE_LAUNCH_DIRECT
}

}  // namespace

namespace qmle {

int launch_direct(const qmle_plan *p, const LoweredOp &op, float2 *states, const float *mats,
                  int batch, hipStream_t stream) {
  const int n = p->n;
  const bool diag = op.flags & LF_DIAG;
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  const int pt = op.t0, pc = op.c0;
  int mode;
  uint64_t items;
  static const bool k1_plain = std::getenv("QMLE_K1_PLAIN") != nullptr;
  if (op.nc == 0) {
    if (pt >= 1) {
      mode = 0;
      items = diag ? chunks : chunks >> 1;
      // dense gate, state >= 2^12 chunks: lane exchange for bits 1..6, 4-row bursts for bits >= 21
      if (!diag && !k1_plain && n >= 14) {
        if (pt <= 6) { mode = 5; items = chunks >> 1; }
        else if (pt >= 21) { mode = 6; items = chunks >> 3; }
      }
    } else { mode = 1; items = chunks; }
  } else {
    if (pc >= 1 && pt >= 1) { mode = 2; items = diag ? chunks >> 1 : chunks >> 2; }
    else if (pt == 0) { mode = 3; items = chunks >> 1; }
    else { mode = 4; items = diag ? chunks : chunks >> 1; }
    // control and target both inside a wave's 1 KiB: one contiguous float4 per lane (mode 7)
    // (controls on bits >= 4 select whole 128-byte lines: mode 2 moves half the state, 0.37 vs 0.65 ms)
    if (!diag && !k1_plain && n >= 14 && pt >= 1 && pt <= 6 && pc >= 0 && pc <= 3) { mode = 7; items = chunks; }
    // bursts of 4 rows per stream (mode 8) where control and target leave a wave's 4 KiB whole.  Which
    // positions gain is measured, not derived (n = 28, tools/k1_block_order.py with QMLE_K1_CTRL_BURST=0 / 9
    // for four control placements, profiles/r04_k1_ctrl_burst.txt): target position >= 21 always (0.35-0.42
    // -> 0.34-0.37 ms), target position 9 .. 11 (0.38-0.41 -> 0.35-0.37), and the neighbouring control below
    // the target from position 17 up (0.36-0.39 -> 0.34-0.35); target positions 12 .. 16 stream best one
    // pair per work item (+5 .. +9 % in bursts).  QMLE_K1_CTRL_BURST=<min target position | 0> overrides.
    if (mode == 2 && !diag && !k1_plain && n >= 16 && pt >= 9 && pc >= 9) {
      bool burst = n >= 24 && (pt >= 21 || pt <= 11 || (pc == pt - 1 && pt >= 17));
      if (const char *e = std::getenv("QMLE_K1_CTRL_BURST")) burst = atoi(e) > 0 && pt >= atoi(e);
      if (burst) {
        mode = 8;
        items = chunks >> 4;
      }
    }
  }
  // CZ / CPhase with control and target on chunk bits: the |11> quarter only (mode 9)
  if (diag && (op.flags & LF_PHASE) && op.nc == 1 && pc >= 1 && pt >= 1 && !k1_plain) { mode = 9; items = chunks >> 2; }
  if (items == 0) items = 1;
  // streaming (non-temporal) accesses once the working set dwarfs the Infinity Cache
  const bool nt = ((size_t)batch << n) * sizeof(float2) >= ((size_t)1 << 30);
  dim3 grid((unsigned)((items + 255) / 256), (unsigned)batch);
  float4 *st = reinterpret_cast<float4 *>(states);
  // A control on bit position 7 or 8 is byte-address bit 10 / 11: the gate touches 2 KiB (1 KiB) on, 2 KiB
  // (1 KiB) off, and with workgroups visiting that half in ascending order the HBM channels behind the L2
  // are loaded unevenly (round 3: even TCC_EA0 requests per channel, uneven DRAM credit stalls; 0.50 / 0.44
  // ms at n = 28 where the neighbouring wires take 0.35 - 0.38).  Spreading the workgroups IN FLIGHT over
  // the whole state -- workgroup i takes block i * 4097 mod grid, ~32 MiB apart -- brings both to 0.38 ms
  // (tools/k1_block_order.py, profiles/r04_k1_block_order.txt); every other control position streams best
  // in ascending order (+2 ... +15 % with any multiplier), so only these two get it.
  // QMLE_K1_BLOCK_MUL=<odd | 0> overrides (read per launch).
  uint32_t blk_mul = 0;
  if (op.nc && (mode == 2 || mode == 9) && (grid.x & (grid.x - 1u)) == 0 && items == (uint64_t)grid.x * 256u) {
    if (!diag && (pc == 7 || pc == 8) && n >= 24 && grid.x > 4097u) blk_mul = 4097u;
    // (round 5, K1 CRZ / CZ / CPhase: the diagonal forms show the same pattern -- CRZ 0.46 ms with the control on
    // position 8, CZ 0.23-0.25 ms with control or target on 7 / 8 where their neighbours take 0.33 / 0.17)
    if (diag && n >= 24 && grid.x > 4097u && (pc == 7 || pc == 8 || (mode == 9 && (pt == 7 || pt == 8)))) blk_mul = 4097u;
    // (the |11> quarter of a pair inside positions 16..18 -- byte-address bits 19..21, 0.5-2 MiB on, the rest of
    // every 4 MiB off -- streams at 0.21-0.26 ms in ascending order where its neighbours take 0.17; workgroups 17
    // blocks apart: 0.19-0.20.  Every other pair is fastest ascending: profiles/r05_k1_cz_order.txt)
    if (mode == 9 && n >= 24 && grid.x > 4097u && pc >= 16 && pc <= 18 && pt >= 16 && pt <= 18) blk_mul = 17u;
    const char *e = std::getenv("QMLE_K1_BLOCK_MUL");
    if (e) blk_mul = atoi(e) > 0 ? ((uint32_t)atoi(e) | 1u) : 0u;
  }
  switch (mode) {
    case 0: launch_direct_mode<0>(diag, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items, blk_mul); break;
    case 1: launch_direct_mode<1>(diag, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items, blk_mul); break;
    case 2: launch_direct_mode<2>(diag, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items, blk_mul); break;
    case 3: launch_direct_mode<3>(diag, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items, blk_mul); break;
    case 5: launch_direct_mode<5>(false, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items, blk_mul); break;
    case 6: launch_direct_mode<6>(false, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items, blk_mul); break;
    case 7: launch_direct_mode<7>(false, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items, blk_mul); break;
    case 8: launch_direct_mode<8>(false, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items, blk_mul); break;
    case 9: launch_direct_mode<9>(true, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items, blk_mul); break;
    default: launch_direct_mode<4>(diag, nt, grid, stream, st, n, pt, pc, mats, p->mat_floats, op.mat_off, items, blk_mul); break;
  }
  HIPCHK(hipGetLastError());
  return QMLE_OK;
}

void launch_init_zero(float2 *states, int n, int batch, hipStream_t stream) {
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  hipLaunchKernelGGL(k_init_zero, dim3(grid_for(chunks, 256), (unsigned)batch), dim3(256), 0, stream,
                     reinterpret_cast<float4 *>(states), n);
}

void launch_diag_all(float2 *states, int n, int batch, const float *marks, const float *angles,
                     int n_slots, int slot, hipStream_t stream) {
  const uint64_t chunks = (uint64_t)1 << (n - 1);
  hipLaunchKernelGGL(k_diag_all, dim3(grid_for(chunks, 256), (unsigned)batch), dim3(256), 0, stream,
                     reinterpret_cast<float4 *>(states), n, marks, angles, n_slots, slot);
}

// zero fill of `count` float4 (grid.x < 2^31 workgroups per launch)
void launch_fill_zero(float2 *states, uint64_t count, hipStream_t stream) {
  for (uint64_t done = 0; done < count;) {
    const uint64_t part = std::min<uint64_t>(count - done, (uint64_t)1 << 38);
    // (QMLE_FILL_NT=1: streaming stores -- the fill by itself is a little slower with them, tools/fill_bench.hip; the
    // pass that reads the zeros next does not run into their write-back.  Read per launch: A/B)
    const char *e = std::getenv("QMLE_FILL_NT");
    if (e ? atoi(e) != 0 : kFillNtDefault)
      hipLaunchKernelGGL(k_fill_zero<true>, dim3((unsigned)((part + 255u) / 256u)), dim3(256), 0, stream,
                         reinterpret_cast<float4 *>(states) + done, part);
    else
      hipLaunchKernelGGL(k_fill_zero<false>, dim3((unsigned)((part + 255u) / 256u)), dim3(256), 0, stream,
                         reinterpret_cast<float4 *>(states) + done, part);
    done += part;
  }
}

}  // namespace qmle
