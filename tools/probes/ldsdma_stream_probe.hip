// LDS-DMA streaming probe (not part of the product): what read rate does a `global_load_lds_dwordx4` ring reach when every workgroup streams a long
// contiguous-per-chunk range, as a function of threads per workgroup, ring depth, bytes per stage, workgroups per CU and the chunk -> workgroup map?
// Build: hipcc --offload-arch=gfx950 -O3 -o ldsdma_stream_probe ldsdma_stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// NW waves; stage = PCW pieces of 1 KB per wave; ST stages; chunk c of workgroup w: interleaved (c * G + w) or contiguous (w * nc + c)
template <int NW, int PCW, int ST, int INTERLEAVE, int CONSUME>
__global__ __launch_bounds__(NW * 64) void kdma(const char* __restrict__ src, long nchunks_total, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STG = NW * PCW * 1024;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long G = gridDim.x, w = blockIdx.x;
  const long nc = (nchunks_total - w + G - 1) / G;   // chunks of this workgroup (both maps: same count up to one)
  const long per = nchunks_total / G;
  int ic = 0, islot = 0;
  auto issue = [&]() {
    const long c = INTERLEAVE ? (long)ic * G + w : w * per + ic;
    const char* base = src + c * STG + wave * (PCW * 1024) + lane * 16;
    char* slot = smem + islot * STG + wave * (PCW * 1024);
#pragma unroll
    for (int i = 0; i < PCW; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + i * 1024), (__attribute__((address_space(3))) void*)(slot + i * 1024), 16, 0, 0);
    ++ic;
    if (++islot == ST) islot = 0;
  };
  const long n = INTERLEAVE ? nc : per;
  unsigned acc = 0;
#pragma unroll
  for (int s = 0; s < ST - 1; ++s)
    if (s < n) issue();
  int cslot = 0;
  for (long c = 0; c < n; ++c) {
    if (n - 1 - c >= ST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * PCW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (c + ST - 1 < n) issue();
    if (CONSUME) {
      const unsigned* p = reinterpret_cast<const unsigned*>(smem + cslot * STG);
#pragma unroll
      for (int i = 0; i < CONSUME; ++i) acc += p[(threadIdx.x + i * NW * 64) & (STG / 4 - 1)];
    }
    if (++cslot == ST) cslot = 0;
  }
  if (acc == 0x12345678u) *sink = acc;
}

// plain loads to registers, U x 16 B per thread per iteration, grid-stride
template <int U>
__global__ __launch_bounds__(256) void kreg(const uint4* __restrict__ a, long n, unsigned* sink) {
  const long stride = (long)gridDim.x * 256 * U;
  unsigned acc = 0;
  for (long i0 = (long)blockIdx.x * 256 * U + threadIdx.x; i0 < n; i0 += stride) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = a[i0 + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

template <int NW, int PCW, int ST, int IL, int CONSUME>
void run(const char* src, long bytes, int wg_per_cu, unsigned* sink) {
  constexpr int STG = NW * PCW * 1024, lds = ST * STG;
  CK(hipFuncSetAttribute((const void*)kdma<NW, PCW, ST, IL, CONSUME>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int g = 256 * wg_per_cu;
  const long nchunks = bytes / STG / g * g;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((kdma<NW, PCW, ST, IL, CONSUME>), dim3(g), dim3(NW * 64), lds, 0, src, nchunks, sink);
  CK(hipEventRecord(e0));
  const int R = 5;
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL((kdma<NW, PCW, ST, IL, CONSUME>), dim3(g), dim3(NW * 64), lds, 0, src, nchunks, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= R;
  printf("dma  waves=%d stage=%3d KB ring=%d (%3d KB/WG) wg/cu=%d %s consume=%2d : %.3f ms  %.2f TB/s\n", NW, STG / 1024, ST, lds / 1024, wg_per_cu, IL ? "interleaved" : "contiguous ",
         CONSUME, ms, (double)nchunks * STG / ms * 1e-9);
}

int main() {
  const long bytes = 3L << 30;
  char* a; unsigned* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&sink, 16)); CK(hipMemset(a, 1, bytes));
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int g : {2048, 4096, 8192}) {
      hipLaunchKernelGGL((kreg<4>), dim3(g), dim3(256), 0, 0, (const uint4*)a, bytes / 16, sink);
      CK(hipEventRecord(e0));
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((kreg<4>), dim3(g), dim3(256), 0, 0, (const uint4*)a, bytes / 16, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
      printf("reg  U=4 grid=%d : %.3f ms  %.2f TB/s\n", g, ms, (double)bytes / ms * 1e-9);
    }
  }
  // the streaming weight-gradient kernel's geometry: 8 waves, 32-KB stages, ring of 4, one workgroup per CU
  run<8, 4, 4, 0, 0>(a, bytes, 1, sink);
  run<8, 4, 4, 1, 0>(a, bytes, 1, sink);
  run<8, 4, 4, 1, 16>(a, bytes, 1, sink);
  run<8, 4, 3, 1, 0>(a, bytes, 1, sink);
  run<8, 4, 2, 1, 0>(a, bytes, 1, sink);
  run<8, 2, 8, 1, 0>(a, bytes, 1, sink);
  run<8, 2, 4, 1, 0>(a, bytes, 2, sink);
  run<8, 1, 8, 1, 0>(a, bytes, 2, sink);
  run<8, 1, 4, 1, 0>(a, bytes, 4, sink);
  run<4, 4, 4, 1, 0>(a, bytes, 2, sink);
  run<4, 4, 4, 1, 0>(a, bytes, 1, sink);
  run<4, 2, 4, 1, 0>(a, bytes, 4, sink);
  run<4, 8, 4, 1, 0>(a, bytes, 1, sink);
  run<4, 4, 8, 1, 0>(a, bytes, 1, sink);
  run<4, 2, 8, 1, 0>(a, bytes, 2, sink);
  run<1, 8, 8, 1, 0>(a, bytes, 1, sink);
  run<1, 8, 8, 1, 0>(a, bytes, 2, sink);
  run<2, 8, 8, 1, 0>(a, bytes, 1, sink);
  run<16, 2, 4, 1, 0>(a, bytes, 1, sink);
  return 0;
}
