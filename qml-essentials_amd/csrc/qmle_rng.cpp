// Host-side parameter sampler (no device code).
//
// The reference draws initial parameters with jax.random.uniform under a threefry key
// (qml_essentials/model.py:687-693); that stream cannot be reproduced without JAX, so
// this build's keys are numpy SeedSequences and its stream is numpy's Philox4x64-10
// (qml-essentials_amd/utils.py).  numpy spends ~9 ns per value there; Expressibility's 2 x 1024
// parameter sets at 12 qubits (73 728 values) cost more host time than the whole GPU call.  This
// file restates the generator -- the Random123 Philox4x64 round function with 10 rounds, numpy's
// counter convention (incremented BEFORE a block is produced, four outputs per block, in order)
// and its double conversion ((x >> 11) * 2^-53) -- so that
//   qmle_philox_uniform_f32(key, n, low, high, out)
// writes exactly numpy.random.Generator(numpy.random.Philox(key=key)).uniform(low, high, n)
// .astype(float32); tests/test_abi_cpu.py compares the two bit for bit.
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

#include "qmle_sv.h"

namespace {

inline void mulhilo(uint64_t a, uint64_t b, uint64_t &hi, uint64_t &lo) {
  const unsigned __int128 p = (unsigned __int128)a * b;
  hi = (uint64_t)(p >> 64);
  lo = (uint64_t)p;
}

}  // namespace

// W blocks in lockstep: a block's ten rounds are a chain of dependent 64 x 64 -> 128-bit multiplies
// (~60 cycles); independent blocks fill the multiplier's pipeline
template <int W>
static inline void philox_blocks(const uint64_t *ctr0, uint64_t k0, uint64_t k1, uint64_t (*out)[4]) {
  const uint64_t M0 = 0xD2E7470EE14C6C93ull, M1 = 0xCA5A826395121157ull;
  const uint64_t W0 = 0x9E3779B97F4A7C15ull, W1 = 0xBB67AE8584CAA73Bull;
  uint64_t c[W][4];
  for (int w = 0; w < W; ++w) {  // ctr0 + w (callers keep the low word from wrapping inside a batch)
    c[w][0] = ctr0[0] + (uint64_t)w;
    c[w][1] = ctr0[1];
    c[w][2] = ctr0[2];
    c[w][3] = ctr0[3];
  }
  for (int r = 0; r < 10; ++r) {
    for (int w = 0; w < W; ++w) {
      uint64_t hi0, lo0, hi1, lo1;
      mulhilo(M0, c[w][0], hi0, lo0);
      mulhilo(M1, c[w][2], hi1, lo1);
      const uint64_t n0 = hi1 ^ c[w][1] ^ k0, n2 = hi0 ^ c[w][3] ^ k1;
      c[w][0] = n0; c[w][1] = lo1; c[w][2] = n2; c[w][3] = lo0;
    }
    k0 += W0;
    k1 += W1;
  }
  for (int w = 0; w < W; ++w)
    for (int j = 0; j < 4; ++j) out[w][j] = c[w][j];
}

// values [0, n) of the stream that starts at block `first_block` (block b = counter b + 1: numpy
// increments the counter before it produces a block)
static void philox_fill(const uint64_t key[2], uint64_t first_block, uint64_t n, double low, double range,
                        float *out) {
#pragma clang fp contract(off)  // numpy evaluates low + range * u in two roundings
  constexpr int W = 3;
  uint64_t ctr[4] = {first_block + 1, 0, 0, 0};  // (< 2^64 blocks: the high words stay zero)
  uint64_t i = 0;
  for (; i + 4 * W <= n; i += 4 * W, ctr[0] += W) {
    uint64_t b[W][4];
    philox_blocks<W>(ctr, key[0], key[1], b);
    for (int w = 0; w < W; ++w)
      for (int j = 0; j < 4; ++j)
        out[i + 4 * w + j] = (float)(low + range * ((double)(b[w][j] >> 11) * (1.0 / 9007199254740992.0)));
  }
  for (; i < n; i += 4, ++ctr[0]) {
    uint64_t b[1][4];
    philox_blocks<1>(ctr, key[0], key[1], b);
    const uint64_t m = n - i < 4 ? n - i : 4;
    for (uint64_t j = 0; j < m; ++j)
      out[i + j] = (float)(low + range * ((double)(b[0][j] >> 11) * (1.0 / 9007199254740992.0)));
  }
}

extern "C" int qmle_philox_uniform_f32(const uint64_t key[2], uint64_t n, double low, double high,
                                       float *out) {
  if (!key || (!out && n > 0) || n > (1ull << 62)) return QMLE_ERR_INVALID_ARG;
  const double range = high - low;
  // counter-based: any block can be produced on its own, so large draws are cut into contiguous
  // ranges of blocks, one per thread.  Only large ones: inside an analysis loop the caller has just
  // waited for the GPU and fresh threads land on sleeping cores -- Expressibility(12 q, 1024 pairs:
  // 73 728 values) 0.82 ms per call with one thread, 0.89-1.47 with two or four (MI355X host; in a
  // hot loop 0.14 / 0.11 ms); 2^20 values: 1.93 ms on one thread, 0.41 on eight.
  uint64_t threads = n / 262144;
  const unsigned hw = std::thread::hardware_concurrency();
  if (threads > 8) threads = 8;
  if (hw && threads > hw) threads = hw;
  if (const char *e = std::getenv("QMLE_RNG_THREADS")) threads = (uint64_t)(atoi(e) > 0 ? atoi(e) : 1);  // (tuning)
  if (threads > (n + 3) / 4) threads = (n + 3) / 4 ? (n + 3) / 4 : 1;
  if (threads <= 1) {
    philox_fill(key, 0, n, low, range, out);
    return QMLE_OK;
  }
  const uint64_t blocks = (n + 3) / 4, per = (blocks + threads - 1) / threads;
  std::vector<std::thread> pool;
  for (uint64_t t = 1; t < threads; ++t) {
    const uint64_t b0 = t * per, b1 = (t + 1) * per < blocks ? (t + 1) * per : blocks;
    if (b0 >= b1) break;
    const uint64_t cnt = (b1 * 4 < n ? b1 * 4 : n) - b0 * 4;
    pool.emplace_back(philox_fill, key, b0, cnt, low, range, out + b0 * 4);
  }
  philox_fill(key, 0, (per * 4 < n ? per * 4 : n), low, range, out);
  for (std::thread &th : pool) th.join();
  return QMLE_OK;
}
