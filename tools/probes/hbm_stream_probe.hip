// HBM streaming probe (not part of the product): what rate does a 2-read + 1-write elementwise pass over 3.1-GB bf16 tensors reach on this
// part, as a function of launch shape, unroll and cache policy?  Build: hipcc --offload-arch=gfx950 -O3 -o hbm_stream_probe hbm_stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U, int NT, int MODE>   // MODE 0: c = f(a, b); 1: read a, b only (sum to a sink); 2: write only; 3: c = f(a)
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ c, long n, u32x4* sink) {
  const long stride = (long)gridDim.x * 256 * U;
  u32x4 acc = {0, 0, 0, 0};
  for (long i0 = (long)blockIdx.x * 256 * U + threadIdx.x; i0 < n; i0 += stride) {
    u32x4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * 256;
      if (MODE != 2) va[u] = NT ? __builtin_nontemporal_load(a + i) : a[i];
      if (MODE == 0 || MODE == 1) vb[u] = NT ? __builtin_nontemporal_load(b + i) : b[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * 256;
      u32x4 r;
      if (MODE == 0) r = va[u] ^ (vb[u] + 1u);
      else if (MODE == 3) r = va[u] + 1u;
      else if (MODE == 2) r = u32x4{(unsigned)i, 1u, 2u, 3u};
      if (MODE == 1) acc += va[u] ^ vb[u];
      else { if (NT) __builtin_nontemporal_store(r, c + i); else c[i] = r; }
    }
  }
  if (MODE == 1 && acc.x == 0x12345678u) *sink = acc;
}

template <int U, int NT, int MODE>
void run(const char* name, u32x4* a, u32x4* b, u32x4* c, long n, int blocks, u32x4* sink) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long per = 256L * U;
  const int g = blocks > 0 ? blocks : (int)((n + per - 1) / per);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<U, NT, MODE>), dim3(g), dim3(256), 0, 0, a, b, c, n, sink);
  CK(hipEventRecord(e0));
  const int R = 10;
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL((k<U, NT, MODE>), dim3(g), dim3(256), 0, 0, a, b, c, n, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= R;
  const double units = MODE == 0 ? 3 : (MODE == 1 ? 2 : (MODE == 2 ? 1 : 2));
  printf("%-28s U=%d nt=%d grid=%7d  %.3f ms  %.2f TB/s\n", name, U, NT, g, ms, units * n * 16 / ms * 1e-9);
}

int main() {
  const long n = 8L * 160 * 160 * 160 * 48 * 2 / 16;   // 16-byte elements of one 8-grid decoder1 tensor (3.15 GB)
  u32x4 *a, *b, *c, *sink;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c, n * 16)); CK(hipMalloc(&sink, 16));
  CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 2, n * 16)); CK(hipMemset(c, 0, n * 16));
#define ALLG(U, NT, MODE, name) run<U, NT, MODE>(name, a, b, c, n, 0, sink); run<U, NT, MODE>(name, a, b, c, n, 2048, sink); run<U, NT, MODE>(name, a, b, c, n, 4096, sink); run<U, NT, MODE>(name, a, b, c, n, 1024, sink);
  ALLG(1, 0, 0, "2r1w") ALLG(2, 0, 0, "2r1w") ALLG(4, 0, 0, "2r1w")
  ALLG(1, 1, 0, "2r1w nt") ALLG(2, 1, 0, "2r1w nt") ALLG(4, 1, 0, "2r1w nt")
  ALLG(2, 0, 1, "2r") ALLG(4, 0, 1, "2r") ALLG(4, 1, 1, "2r nt")
  ALLG(2, 0, 2, "1w") ALLG(2, 1, 2, "1w nt")
  ALLG(2, 0, 3, "1r1w") ALLG(2, 1, 3, "1r1w nt") ALLG(4, 1, 3, "1r1w nt")
  return 0;
}
