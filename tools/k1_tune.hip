// Stand-alone tuning harness for the HBM-streaming 1-qubit gate kernel (K1, n = 28).
// Variants: UNROLL (independent pair updates in flight per thread), non-temporal
// accesses, block size, exact vs persistent grid.  Prints ms and algorithmic GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Mat2 { float2 m00, m01, m10, m11; };
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x*b.x - a.y*b.y, a.x*b.y + a.y*b.x); }
__device__ __forceinline__ float2 cfma(float2 a, float2 b, float2 c) { return make_float2(fmaf(a.x,b.x,fmaf(-a.y,b.y,c.x)), fmaf(a.x,b.y,fmaf(a.y,b.x,c.y))); }
__device__ __forceinline__ void apply2(const Mat2& m, float2& a0, float2& a1) {
  float2 b0 = cfma(m.m01, a1, cmul(m.m00, a0)); float2 b1 = cfma(m.m11, a1, cmul(m.m10, a0)); a0 = b0; a1 = b1; }
__device__ __forceinline__ uint64_t ins0(uint64_t i, int p) { return ((i >> p) << (p+1)) | (i & ((1ull<<p)-1)); }

typedef float vf4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 ld(const float4* p) {
  if (NT) { vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
  else return *p; }
template <bool NT> __device__ __forceinline__ void st(float4* p, float4 v) {
  if (NT) { vf4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<vf4*>(p)); }
  else *p = v; }

// MODE 0 of k_direct_1q: target chunk-bit q = pt-1 >= 0
template <int UNROLL, bool NT, int BS>
__global__ void __launch_bounds__(BS) k_rx(float4* __restrict__ s, int q, Mat2 m, uint64_t items) {
  const uint64_t stride = (uint64_t)gridDim.x * BS;
  uint64_t k = (uint64_t)blockIdx.x * BS + threadIdx.x;
  for (; k + (UNROLL-1)*stride < items; k += UNROLL*stride) {
    float4 v0[UNROLL], v1[UNROLL]; uint64_t c0[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { c0[u] = ins0(k + u*stride, q); v0[u] = ld<NT>(s + c0[u]); v1[u] = ld<NT>(s + (c0[u] | (1ull<<q))); }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      float2 a0 = make_float2(v0[u].x, v0[u].y), a1 = make_float2(v1[u].x, v1[u].y);
      float2 b0 = make_float2(v0[u].z, v0[u].w), b1 = make_float2(v1[u].z, v1[u].w);
      apply2(m, a0, a1); apply2(m, b0, b1);
      st<NT>(s + c0[u], make_float4(a0.x,a0.y,b0.x,b0.y));
      st<NT>(s + (c0[u] | (1ull<<q)), make_float4(a1.x,a1.y,b1.x,b1.y));
    }
  }
  for (; k < items; k += stride) {
    uint64_t c = ins0(k, q); float4 x = s[c], y = s[c | (1ull<<q)];
    float2 a0 = make_float2(x.x,x.y), a1 = make_float2(y.x,y.y), b0 = make_float2(x.z,x.w), b1 = make_float2(y.z,y.w);
    apply2(m,a0,a1); apply2(m,b0,b1);
    s[c] = make_float4(a0.x,a0.y,b0.x,b0.y); s[c | (1ull<<q)] = make_float4(a1.x,a1.y,b1.x,b1.y);
  }
}

// copy-scale reference: read + write every chunk (the in-place streaming ceiling)
template <bool NT, int BS>
__global__ void __launch_bounds__(BS) k_scale(float4* __restrict__ s, uint64_t chunks) {
  const uint64_t stride = (uint64_t)gridDim.x * BS;
  for (uint64_t k = (uint64_t)blockIdx.x * BS + threadIdx.x; k < chunks; k += stride) {
    float4 v = ld<NT>(s + k); v.x *= 1.0001f; st<NT>(s + k, v); }
}


// Round 2: a thread takes U ADJACENT wave rows of each stream (U KiB contiguous per wave and
// stream), all loads of stream 0 first, then stream 1: fewer alternations between the two DRAM
// rows a bank sees.  Exact grid.
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_rx_rows(float4* __restrict__ s, int q, Mat2 m, uint64_t items) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t row0 = ((uint64_t)blockIdx.x * 4u + wave) * U;
  float4 v0[U], v1[U]; uint64_t c0[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { c0[u] = ins0((row0 + u) * 64u + lane, q); v0[u] = ld<NT>(s + c0[u]); }
#pragma unroll
  for (int u = 0; u < U; ++u) v1[u] = ld<NT>(s + (c0[u] | (1ull << q)));
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float2 a0 = make_float2(v0[u].x, v0[u].y), a1 = make_float2(v1[u].x, v1[u].y);
    float2 b0 = make_float2(v0[u].z, v0[u].w), b1 = make_float2(v1[u].z, v1[u].w);
    apply2(m, a0, a1); apply2(m, b0, b1);
    v0[u] = make_float4(a0.x,a0.y,b0.x,b0.y); v1[u] = make_float4(a1.x,a1.y,b1.x,b1.y);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) st<NT>(s + c0[u], v0[u]);
#pragma unroll
  for (int u = 0; u < U; ++u) st<NT>(s + (c0[u] | (1ull << q)), v1[u]);
}

// Round 2, q < 6: the partner chunk sits in lane ^ (1 << q) of the same wave.  Every lane loads
// and stores ONE contiguous float4 (fully coalesced, like the diagonal gate) and fetches the
// partner's through the cross-lane path; it computes only its own half of the pair.
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_rx_lane(float4* __restrict__ s, int q, Mat2 m, uint64_t chunks) {
  const uint64_t k0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x);
  const uint64_t stride = (uint64_t)gridDim.x * 256u;
  const bool up = (threadIdx.x >> q) & 1u;
  const float2 ms = up ? m.m11 : m.m00, mo = up ? m.m10 : m.m01;
  float4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = ld<NT>(s + k0 + u * stride);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float4 o;
    o.x = __shfl_xor(v[u].x, 1 << q, 64); o.y = __shfl_xor(v[u].y, 1 << q, 64);
    o.z = __shfl_xor(v[u].z, 1 << q, 64); o.w = __shfl_xor(v[u].w, 1 << q, 64);
    const float2 a = cfma(mo, make_float2(o.x, o.y), cmul(ms, make_float2(v[u].x, v[u].y)));
    const float2 b = cfma(mo, make_float2(o.z, o.w), cmul(ms, make_float2(v[u].z, v[u].w)));
    st<NT>(s + k0 + u * stride, make_float4(a.x, a.y, b.x, b.y));
  }
}

template <typename F> float time_ms(F f, int reps = 7) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
  std::sort(t.begin(), t.end()); return t[t.size()/2];
}

template <int UNROLL, bool NT, int BS>
void run(float4* d, int n, int q, unsigned grid_cap, const char* tag) {
  const uint64_t chunks = 1ull << (n-1), items = chunks >> 1;
  Mat2 m{{0.8f,0.f},{0.f,-0.6f},{0.f,-0.6f},{0.8f,0.f}};
  uint64_t g = (items + (uint64_t)BS*UNROLL - 1) / ((uint64_t)BS*UNROLL);
  if (grid_cap && g > grid_cap) g = grid_cap;
  float ms = time_ms([&]{ hipLaunchKernelGGL((k_rx<UNROLL,NT,BS>), dim3((unsigned)g), dim3(BS), 0, 0, d, q, m, items); });
  printf("%-28s q=%2d grid=%8llu  %7.4f ms  %7.1f GB/s\n", tag, q, (unsigned long long)g, ms, 16.0*(1ull<<n)/ms/1e6);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 28;
  const uint64_t D = 1ull << n;
  float4* d; CK(hipMalloc(&d, D*8)); CK(hipMemset(d, 0, D*8));
  {
    const uint64_t chunks = D/2;
    float ms = time_ms([&]{ hipLaunchKernelGGL((k_scale<false,256>), dim3(8192), dim3(256), 0, 0, d, chunks); });
    printf("scale in place (default)      %7.4f ms %7.1f GB/s\n", ms, 16.0*D/ms/1e6);
    ms = time_ms([&]{ hipLaunchKernelGGL((k_scale<true,256>), dim3(8192), dim3(256), 0, 0, d, chunks); });
    printf("scale in place (nontemporal)  %7.4f ms %7.1f GB/s\n", ms, 16.0*D/ms/1e6);
    ms = time_ms([&]{ hipLaunchKernelGGL((k_scale<false,256>), dim3((unsigned)(chunks/256)), dim3(256), 0, 0, d, chunks); });
    printf("scale in place (exact grid)   %7.4f ms %7.1f GB/s\n", ms, 16.0*D/ms/1e6);
    ms = time_ms([&]{ hipLaunchKernelGGL((k_scale<true,256>), dim3((unsigned)(chunks/256)), dim3(256), 0, 0, d, chunks); });
    printf("scale in place (exact, nt)    %7.4f ms %7.1f GB/s\n", ms, 16.0*D/ms/1e6);
    ms = time_ms([&]{ hipLaunchKernelGGL((k_scale<false,1024>), dim3(2048), dim3(1024), 0, 0, d, chunks); });
    printf("scale in place (bs1024 g2048) %7.4f ms %7.1f GB/s\n", ms, 16.0*D/ms/1e6);
  }
  if (argc > 2) {  // round-2 sweep: every stride, three variants
    Mat2 m{{0.8f,0.f},{0.f,-0.6f},{0.f,-0.6f},{0.8f,0.f}};
    const uint64_t chunks = 1ull << (n-1), items = chunks >> 1;
    for (int q = 0; q < n - 1; ++q) {
      float t1 = time_ms([&]{ hipLaunchKernelGGL((k_rx<1,true,256>), dim3((unsigned)(items/256)), dim3(256), 0, 0, d, q, m, items); });
      float t2 = time_ms([&]{ hipLaunchKernelGGL((k_rx_rows<2,true>), dim3((unsigned)(items/512)), dim3(256), 0, 0, d, q, m, items); });
      float t4 = time_ms([&]{ hipLaunchKernelGGL((k_rx_rows<4,true>), dim3((unsigned)(items/1024)), dim3(256), 0, 0, d, q, m, items); });
      float t8 = time_ms([&]{ hipLaunchKernelGGL((k_rx_rows<8,true>), dim3((unsigned)(items/2048)), dim3(256), 0, 0, d, q, m, items); });
      float tl = -1.f, tl2 = -1.f;
      if (q < 6) {
        tl = time_ms([&]{ hipLaunchKernelGGL((k_rx_lane<1,true>), dim3((unsigned)(chunks/256)), dim3(256), 0, 0, d, q, m, chunks); });
        tl2 = time_ms([&]{ hipLaunchKernelGGL((k_rx_lane<2,true>), dim3((unsigned)(chunks/512)), dim3(256), 0, 0, d, q, m, chunks); });
      }
      printf("q=%2d (bit %2d)  pair/thread %.4f  rows2 %.4f  rows4 %.4f  rows8 %.4f  lane %.4f lane2 %.4f ms\n", q, q + 1, t1, t2, t4, t8, tl, tl2);
      fflush(stdout);
    }
    return 0;
  }
  for (int q : {26, 13, 0}) {
    run<1,false,256>(d, n, q, 8192, "u1 bs256 cap8192");
    run<2,false,256>(d, n, q, 8192, "u2 bs256 cap8192");
    run<4,false,256>(d, n, q, 8192, "u4 bs256 cap8192");
    run<1,false,256>(d, n, q, 0, "u1 bs256 exact");
    run<2,false,256>(d, n, q, 0, "u2 bs256 exact");
    run<4,false,256>(d, n, q, 0, "u4 bs256 exact");
    run<1,true,256>(d, n, q, 0, "u1 bs256 exact nt");
    run<2,true,256>(d, n, q, 0, "u2 bs256 exact nt");
    run<4,true,256>(d, n, q, 0, "u4 bs256 exact nt");
    run<2,true,256>(d, n, q, 4096, "u2 bs256 cap4096 nt");
    run<2,false,512>(d, n, q, 0, "u2 bs512 exact");
    run<2,false,1024>(d, n, q, 2048, "u2 bs1024 cap2048");
    run<4,true,1024>(d, n, q, 1024, "u4 bs1024 cap1024 nt");
  }
  return 0;
}
